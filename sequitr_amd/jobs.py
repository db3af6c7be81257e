"""The `jobs` plugin module: job functions with the signature ``func(params, options)`` that
``JobWrapper.__call__`` imports and calls (sequitr/worker.py:56-57, 208-215).  The
reference names this module in its example job file but does not ship it; these
functions are the drop-in bodies that route the per-tile hot path to the MI355X.

Job-file example::

    [job]
    ID = 467e3c03
    user = Alan
    priority = 99
    device = GPU
    module = sequitr_amd.jobs
    func = SERVER_segment
    params = {'input': '/data/tiles.npy', 'shape': (512, 512), 'num_outputs': 2}
    options = {'gpu': 0, 'save_logits': True}

There is no CPU back end: ``device = CPU`` jobs fail loudly (the exception is logged by the
worker's exception_logger, as every job error is).
"""
import json
import logging
import os
import time

import numpy as np

logger = logging.getLogger('worker_process')

NET_KEYS = ('name', 'filters', 'dropout', 'num_inputs', 'num_outputs', 'shape', 'bridge', 'kernel', 'seed', 'dtype',
            'batch_norm', 'up_kernel')


def _resolve_device(params, options):
    """job.device / options['gpu'] / LOCAL_RANK -> torch device string."""
    dev = str(params.get('device', 'GPU'))
    if dev.upper() == 'CPU':
        raise RuntimeError("sequitr_amd has no CPU back end: submit the job with device = GPU")
    if dev.lower().startswith('cuda'):
        return dev
    idx = options.get('gpu', os.environ.get('LOCAL_RANK', 0))
    return 'cuda:%d' % int(idx)


def _load_tiles(params):
    src = params.get('input')
    if isinstance(src, np.ndarray):
        x = src
    elif isinstance(src, str) and src.endswith('.npy'):
        x = np.load(src, mmap_mode='r', allow_pickle=False)
    elif isinstance(src, dict) and src.get('synthetic'):
        s = src
        x = np.random.default_rng(s.get('seed', 0)).standard_normal(
            (s.get('tiles', 1),) + tuple(params.get('shape', (512, 512))) + (params.get('num_inputs', 1),)
        ).astype(np.float32)
    else:
        raise ValueError("params['input'] must be a .npy path, an ndarray or {'synthetic': True, ...}")
    if x.ndim == 2:
        x = x[np.newaxis, ..., np.newaxis]
    elif x.ndim == 3:
        x = x[..., np.newaxis]
    return x


def _net_params(params, device):
    p = {k: params[k] for k in NET_KEYS if k in params}
    p['device'] = device
    return p


def SERVER_segment(params, options):
    """Segment a stack of tiles: writes ``mask.npy`` (uint8 class labels, N x H x W) and,
    with options['save_logits'], ``logits.npy`` into params['output'], plus ``segment.json``
    with timing.  params: input, shape, num_inputs, num_outputs, filters, bridge, model
    (numbered model dir or name to warm-start from; else seeded initial weights), pipeline
    (ImagePipeline JSON applied to every tile on the host), batch (tiles per launch batch).
    options: gpu, save_logits, centroids, io_threads (host staging threads, default 4).

    ``segment.json``: ``seconds`` / ``mpixels_per_s`` cover the stream over the whole stack (host tiles in, host
    masks out, PCIe both ways included); ``setup_seconds`` is what comes before it once per job (weights, pinned
    staging buffers, one warm-up batch) and ``mpixels_per_s_with_setup`` the rate with it counted.
    """
    import torch
    from .networks.unet import UNet2D
    from . import utils
    from .pipeline import ImagePipeline
    from .frontend import TileStreamer

    device = _resolve_device(params, options)
    torch.cuda.set_device(torch.device(device))
    out_dir = params['output']
    x = _load_tiles(params)
    N = x.shape[0]
    net_p = _net_params(params, device)
    net_p.setdefault('shape', tuple(x.shape[1:3]))
    net = UNet2D(net_p, 'infer')
    model = params.get('model')
    if model:
        model_dir = model if os.path.isdir(model) else utils.get_latest_model_dir(
            os.path.join(utils.core.TensorflowConfiguration.MODELDIR, model))
        if model_dir is None:
            raise IOError('No saved model found for {0}'.format(model))
        net.load_state_dict(utils.load_model_weights(model_dir))
        logger.info('Loaded weights from {0:s}'.format(model_dir))
    else:
        net.initialize()

    pipe = ImagePipeline.load(params['pipeline']) if params.get('pipeline') else None
    batch = int(params.get('batch', 32))
    want_logits = bool(options.get('save_logits'))
    writer, frames_out = None, {}
    on_batch = None
    if options.get('centroids'):                               # the reference's next step: utils.CentroidWriter
        from .centroids import CentroidWriter, mask_centroids
        writer = CentroidWriter(os.path.join(out_dir, 'tracks.hdf5'))

        def on_batch(first, m):                                # centroids straight from the mask in HBM
            for k, coords in enumerate(mask_centroids(m)):
                coords[:, 0] = first + k                       # frame index within the whole stack
                frames_out[first + k] = coords

    # the streamed data path (frontend.TileStreamer): staging, H2D, the network and D2H of consecutive batches
    # overlap on three streams; set-up (pinned buffers, first-launch costs) is timed apart from the stream itself
    t_setup = time.time()
    streamer = TileStreamer(net, batch=batch, want_logits=want_logits, workers=int(options.get('io_threads', 4)))
    streamer.warm_up(tuple(x.shape[1:]))
    # zeros, not empty: the pages are touched here, in the set-up time -- first-touch faults of a 268 MB array inside the
    # stream are ~15 ms of a 180 ms pass (the download threads write it while the GPU works)
    masks = np.zeros(x.shape[:3], np.uint8)
    logits = np.zeros(x.shape[:3] + (net.n_outputs,), np.float32) if want_logits else None
    t0 = time.time()
    streamer.run(x, out_masks=masks, out_logits=logits, pipe=(lambda t: pipe(t)) if pipe is not None else None,
                 on_batch=on_batch)
    dt = time.time() - t0
    n_objects = 0
    if writer is not None:
        for k in sorted(frames_out):
            writer.add_frame(k, frames_out[k])
            n_objects += len(frames_out[k])
        writer.close()
    np.save(os.path.join(out_dir, 'mask.npy'), masks)
    if logits is not None:
        np.save(os.path.join(out_dir, 'logits.npy'), logits)
    pixels = N * x.shape[1] * x.shape[2]
    info = {'tiles': int(N), 'shape': [int(s) for s in x.shape[1:3]], 'seconds': dt, 'setup_seconds': t0 - t_setup,
            'mpixels_per_s': float(pixels / max(dt, 1e-9) / 1e6),
            'mpixels_per_s_with_setup': float(pixels / max(dt + t0 - t_setup, 1e-9) / 1e6),
            'batch': batch, 'streamed': True, 'device': device}
    if writer is not None:
        info['centroids'] = {'file': os.path.basename(writer.filename), 'objects': int(n_objects)}
    with open(os.path.join(out_dir, 'segment.json'), 'w') as f:
        json.dump(info, f, indent=2)
    logger.info('Segmented {tiles} tiles in {seconds:.3f}s on {device}'.format(**info))
    return info


def SERVER_segment_frames(params, options):
    """Segment whole camera frames (larger than the network tile): params['input'] = an Octopus stream stem
    (sequitr/dataio/octopus.py), a .npy of (F,H,W) uint8/uint16/float32 frames, or an ndarray.  Raw frames
    cross PCIe; ImageNorm, tiling, the U-Net and stitching run on the GPU (sequitr_amd/frontend.py).  Writes
    ``mask.npy`` (F,H,W) uint8, ``segment.json`` and, with options['centroids'], the centroid file.
    params: shape (tile, default (512,512)), margin, frames_per_batch, model / filters / ... as SERVER_segment."""
    import torch
    from .networks.unet import UNet2D
    from . import utils
    from .frontend import segment_frames
    from .dataio import OctopusData

    device = _resolve_device(params, options)
    out_dir = params['output']
    src = params.get('input')
    if isinstance(src, str) and not src.endswith('.npy'):
        frames = OctopusData(src, timeout=params.get('timeout', 60))
        F, (H, W) = len(frames), frames.framesize
    else:
        frames = np.load(src, mmap_mode='r', allow_pickle=False) if isinstance(src, str) else np.asarray(src)
        F, H, W = frames.shape
    net_p = _net_params(params, device)
    net_p.setdefault('shape', (512, 512))
    tile = int(net_p['shape'][0])
    net = UNet2D(net_p, 'infer')
    model = params.get('model')
    if model:
        model_dir = model if os.path.isdir(model) else utils.get_latest_model_dir(
            os.path.join(utils.core.TensorflowConfiguration.MODELDIR, model))
        if model_dir is None:
            raise IOError('No saved model found for {0}'.format(model))
        net.load_state_dict(utils.load_model_weights(model_dir))
    else:
        net.initialize()
    per_frame = {}
    want_centroids = bool(options.get('centroids'))
    masks = np.empty((F, H, W), np.uint8) if want_centroids else None

    def sink(first, m):                                        # centroids need the masks while they are in HBM
        from .centroids import mask_centroids
        masks[first:first + m.shape[0]] = m.cpu().numpy()
        for k, coords in enumerate(mask_centroids(m)):
            coords[:, 0] = first + k
            per_frame[first + k] = coords

    t0 = time.time()
    # without centroids the masks come back through segment_frames' own double-buffered download (batch i-1 drains
    # while batch i runs); with them every batch is visited on the device first
    out = segment_frames(net, frames, tile=tile, margin=int(params.get('margin', 32)),
                         frames_per_batch=int(params.get('frames_per_batch', 4)), on_masks=sink if want_centroids else None)
    if not want_centroids:
        masks = out
    torch.cuda.synchronize()
    dt = time.time() - t0
    np.save(os.path.join(out_dir, 'mask.npy'), masks)
    info = {'frames': int(F), 'shape': [int(H), int(W)], 'tile': tile, 'seconds': dt,
            'mpixels_per_s': float(F * H * W / max(dt, 1e-9) / 1e6), 'device': device}
    if options.get('centroids'):
        from .centroids import CentroidWriter
        with CentroidWriter(os.path.join(out_dir, 'tracks.hdf5')) as cw:
            for k in sorted(per_frame):
                cw.add_frame(k, per_frame[k])
        info['centroids'] = {'file': os.path.basename(cw.filename), 'objects': int(sum(len(v) for v in per_frame.values()))}
    with open(os.path.join(out_dir, 'segment.json'), 'w') as f:
        json.dump(info, f, indent=2)
    logger.info('Segmented {frames} frames in {seconds:.3f}s on {device}'.format(**info))
    return info


def SERVER_test(params, options):
    """Plumbing check (the reference's commented-out SERVER_test, worker.py:300-302):
    writes the params it was called with into the output folder."""
    with open(os.path.join(params['output'], 'test.json'), 'w') as f:
        json.dump({'params': {k: repr(v) for k, v in params.items()},
                   'options': {k: repr(v) for k, v in options.items()}}, f, indent=2)


def _onehot(labels, num_outputs):
    """class-index labels (N,H,W) -> one-hot uint8 (N,H,W,num_outputs); classes >= num_outputs get an
    all-zero row, as tr_augment's ``concat(...)[..., :outputs]`` does (sequitr/networks/unet.py:396-398)."""
    labels = np.asarray(labels)
    if labels.ndim == 4:
        return np.ascontiguousarray(labels[..., :num_outputs], dtype=np.uint8)
    return np.stack([(labels == c) for c in range(num_outputs)], -1).astype(np.uint8)


def SERVER_train(params, options):
    """Train the U-Net on a stack of tiles with the weight-map-weighted softmax cross-entropy.

    params: images (.npy (N,H,W[,C]) float), labels (.npy (N,H,W) class indices or one-hot), weights
    (.npy (N,H,W[,1]); when absent computed with ImageWeightMap(w0, sigma), sequitr/pipeline.py:455-479, on
    the GPU -- sq_weightmap_edt_f32 -- and kept there), dtype ('f32' | 'bf16' activations), plus the
    NetConfiguration keys (name, shape, num_outputs, learning_rate, num_epochs, batch_size, dropout, filters,
    bridge, warm_start ...) and warmup_steps (linear learning-rate warm-up, HISTORY.md section 8).
    Deviation from the reference's defaults: without params['learning_rate'] the step uses train.DEFAULT_LEARNING_RATE
    (0.003) ramped over train.DEFAULT_WARMUP_STEPS (40), not NetConfiguration's 0.01 (sequitr/utils.py:289), which
    diverges on this net under Adam; the values used are written to net.config and train.json.
    options: gpu, max_steps, graph (default True: the step is captured once and replayed as hipGraphs).

    The data path of a step never leaves the device: tiles, one-hot labels and weight maps are uploaded ONCE and stay
    in HBM (2.6 MB per 512x512 tile against 288 GB; a stack above params['resident_gib'], default 64, is staged batch by
    batch through pinned memory instead), an epoch's permutation is one index tensor, a batch is an index_select
    straight into the captured step's static input buffers, the loss of every step lands in a device-side log that
    is read back once per epoch (no per-step .item(): the host runs ahead of the GPU).  Under torchrun
    (WORLD_SIZE > 1) the tiles shard across ranks and gradients are all-reduced over RCCL once per step, between
    the two graphs.  Rank 0 saves ``weights.npz`` + ``net.config`` into the next numbered folder of MODELDIR/<name>/
    (sequitr/utils.py:143-223 layout) and ``train.json`` (losses, ms_per_step) into params['output'].
    """
    import torch
    from . import utils
    from .parallel import epoch_schedule
    from .weightmap import device_weightmaps
    from .train import UNetTrainer

    device = _resolve_device(params, options)
    torch.cuda.set_device(torch.device(device))
    world, rank = int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('RANK', 0))
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            backend = os.environ.get('SQ_DIST_BACKEND', 'nccl')    # 'gloo': two ranks on one card (tests)
            if backend == 'nccl':
                dist.init_process_group('nccl', device_id=torch.device(device))
            else:
                dist.init_process_group(backend)

    cfg_keys = ('name', 'shape', 'num_inputs', 'num_outputs', 'num_epochs', 'learning_rate', 'warm_start', 'dropout')
    config = utils.NetConfiguration.from_params({k: params[k] for k in cfg_keys if k in params})
    x = np.load(params['images'], mmap_mode='r', allow_pickle=False)
    if x.ndim == 3:
        x = x[..., np.newaxis]
    onehot = _onehot(np.load(params['labels'], allow_pickle=False), config.num_outputs)
    if params.get('weights'):
        wmap = np.load(params['weights'], allow_pickle=False).reshape(onehot.shape[:3] + (1,)).astype(np.float32)
    else:                                                      # EDT weight maps on the device, left in HBM
        fg = onehot[..., 1:].sum(-1)
        wmap = device_weightmaps(fg, params.get('w0', 10.), params.get('sigma', 5.), device=device)

    net_p = _net_params(params, device)
    net_p.setdefault('shape', tuple(x.shape[1:3]))
    net_p['dropout'] = float(params.get('dropout', 0.4))
    # learning_rate: the job's own value when it gives one; otherwise the trainer's default, NOT NetConfiguration's 0.01
    # (sequitr/utils.py:289), which diverges on the 5-level net under Adam (train.DEFAULT_LEARNING_RATE, HISTORY.md section 8)
    trainer = UNetTrainer(net_p, learning_rate=params.get('learning_rate'), warmup_steps=params.get('warmup_steps'))
    # net.config must record the hyper-parameters that were USED (it is what a warm start or an audit reads): the
    # trainer's learning rate and warm-up, not NetConfiguration's untouched defaults
    config.learning_rate = trainer.lr
    config.warmup_steps = trainer.warmup_steps
    if config.warm_start:
        latest = config.warm_start_from()
        if latest:
            trainer.load_state_dict(utils.load_model_weights(latest))
            logger.info('Warm start from {0:s}'.format(latest))

    # Every rank must issue the SAME number of optimiser steps (each one is a gradient all-reduce), so the step count
    # comes from rank-independent quantities only and every step is a full batch: one seeded permutation of the whole
    # stack per epoch, cut into world x steps_per_epoch x batch indices; the remainder of the epoch is dropped.
    n_items = int(x.shape[0])
    order_fn, steps_per_epoch = epoch_schedule(n_items, int(params.get('batch_size', 16)), world)
    batch = order_fn.batch
    if order_fn.dropped:
        logger.info('{0} of {1} tiles are left out of every epoch ({2} ranks x {3} steps x batch {4}; a different '
                    'remainder each epoch)'.format(order_fn.dropped, n_items, world, steps_per_epoch, batch))
    epochs = int(params.get('num_epochs', config.num_epochs))
    max_steps = options.get('max_steps')
    total_steps = epochs * steps_per_epoch if not max_steps else min(int(max_steps), epochs * steps_per_epoch)

    per_tile = int(np.prod(x.shape[1:])) * 4 + int(np.prod(onehot.shape[1:])) + int(np.prod(onehot.shape[1:3])) * 4
    resident = n_items * per_tile <= float(params.get('resident_gib', 64)) * 2 ** 30
    dev = torch.device(device)
    if resident:                                               # the whole stack lives in HBM for the whole job
        x_dev = torch.from_numpy(np.array(x, dtype=np.float32, order='C')).to(dev)   # a copy: x is a read-only memmap
        y_dev = torch.from_numpy(np.ascontiguousarray(onehot)).to(dev)
        w_dev = wmap if isinstance(wmap, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(wmap)).to(dev)
    bufs = [torch.empty((batch,) + tuple(x.shape[1:]), dtype=torch.float32, device=dev),
            torch.empty((batch,) + tuple(onehot.shape[1:]), dtype=torch.uint8, device=dev),
            torch.empty((batch,) + tuple(onehot.shape[1:3]) + (1,), dtype=torch.float32, device=dev)]

    def load_batch(idx_dev, idx_host):
        """fill the step's static input buffers with the tiles `idx` (device gather, or pinned staging)"""
        sx, sy, sw = bufs
        if resident:
            torch.index_select(x_dev, 0, idx_dev, out=sx)
            torch.index_select(y_dev, 0, idx_dev, out=sy)
            torch.index_select(w_dev, 0, idx_dev, out=sw)
        else:
            ih = np.asarray(idx_host)                            # the permutation's own order: the same batch as the resident path
            sx.copy_(torch.from_numpy(np.ascontiguousarray(x[ih], dtype=np.float32)).pin_memory(), non_blocking=True)
            sy.copy_(torch.from_numpy(np.ascontiguousarray(onehot[ih])).pin_memory(), non_blocking=True)
            if isinstance(wmap, torch.Tensor):
                torch.index_select(wmap, 0, torch.from_numpy(ih).to(dev), out=sw)
            else:
                sw.copy_(torch.from_numpy(np.ascontiguousarray(wmap[ih])).pin_memory(), non_blocking=True)

    use_graph = bool(options.get('graph', True))
    loss_log = torch.zeros(max(total_steps, 1), dtype=torch.float32, device=dev)
    losses, done = [], 0
    t_start = t_steady = time.time()
    steady_from = 0
    for epoch in range(epochs):
        if done >= total_steps:
            break
        order = order_fn(epoch, rank)
        order_dev = torch.from_numpy(np.ascontiguousarray(order)).to(dev) if resident else None
        first = done
        for s in range(steps_per_epoch):
            if done >= total_steps:
                break
            sl = slice(s * batch, (s + 1) * batch)
            load_batch(order_dev[sl] if resident else None, order[sl])
            if use_graph and done == 0:
                # the first step runs eagerly (it warms every kernel up and sizes the workspaces), then the step is
                # captured: (zero, forward, loss, backward) + (Adam), all-reduce between them; later batches are
                # gathered straight into the capture's static buffers, so step() has nothing to copy
                trainer.capture(*bufs, warmup=1)
                bufs[:] = trainer.static_inputs
                loss_log[0].copy_(trainer.last_loss)
                torch.cuda.synchronize()
                t_steady, steady_from = time.time(), 1         # ms_per_step is the replayed steady state
            else:
                loss_log[done].copy_(trainer.step(*bufs))
            done += 1
        losses.extend(float(v) for v in loss_log[first:done].cpu().numpy())      # ONE read-back per epoch
    torch.cuda.synchronize()
    t_end = time.time()
    steady = done - steady_from
    info = {'steps': done, 'first_loss': losses[0], 'last_loss': losses[-1], 'seconds': t_end - t_start,
            'ms_per_step': (t_end - t_steady) * 1e3 / steady if steady > 0 else None,
            'steady_steps': steady, 'batch_size': batch, 'tiles': n_items, 'resident': bool(resident),
            'graph': use_graph, 'dtype': str(net_p.get('dtype', 'f32')), 'warmup_steps': trainer.warmup_steps,
            'learning_rate': trainer.lr, 'world': world, 'device': device}
    # replica check: data-parallel replicas apply the same all-reduced gradient to the same weights, so their
    # parameters must agree bit for bit; two f64 sums of the flat parameter bucket are compared across the ranks
    flat = trainer.pbucket.flat.double()
    chk = torch.stack([flat.sum(), flat.abs().sum()])
    info['param_checksum'] = [float(v) for v in chk.cpu()]
    if world > 1:
        import torch.distributed as dist
        hi, lo = chk.clone(), chk.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        info['replicas_identical'] = bool(torch.equal(hi, lo))
        if not info['replicas_identical']:
            logger.error('data-parallel replicas have diverged: parameter checksums span {0} .. {1}'.format(
                [float(v) for v in lo.cpu()], [float(v) for v in hi.cpu()]))
    if rank == 0:
        info['model_dir'] = utils.save_model(trainer.state_dict(), config)
        with open(os.path.join(params['output'], 'train.json'), 'w') as f:
            json.dump(dict(info, losses=losses), f, indent=2)
        logger.info('Trained {steps} steps, loss {first_loss:.4f} -> {last_loss:.4f}, saved {model_dir}'.format(**info))
    return info
