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

NET_KEYS = ('name', 'filters', 'dropout', 'num_inputs', 'num_outputs', 'shape', 'bridge', 'kernel', 'seed')


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
    """
    import torch
    from .networks.unet import UNet2D
    from . import utils
    from .pipeline import ImagePipeline

    device = _resolve_device(params, options)
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
    masks = np.empty(x.shape[:3], np.uint8)
    logits = np.empty(x.shape[:3] + (net.n_outputs,), np.float32) if options.get('save_logits') else None
    t0 = time.time()
    for i in range(0, N, batch):
        xb = np.ascontiguousarray(x[i:i + batch], dtype=np.float32)
        if pipe is not None:
            xb = np.stack([pipe(t.copy()) for t in xb]).astype(np.float32)
        m = net.predict(xb)
        masks[i:i + batch] = m.cpu().numpy()
        if logits is not None:
            logits[i:i + batch] = net.logits().cpu().numpy()
    torch.cuda.synchronize()
    dt = time.time() - t0
    np.save(os.path.join(out_dir, 'mask.npy'), masks)
    if logits is not None:
        np.save(os.path.join(out_dir, 'logits.npy'), logits)
    info = {'tiles': int(N), 'shape': [int(s) for s in x.shape[1:3]], 'seconds': dt,
            'mpixels_per_s': float(N * x.shape[1] * x.shape[2] / max(dt, 1e-9) / 1e6), 'device': device}
    with open(os.path.join(out_dir, 'segment.json'), 'w') as f:
        json.dump(info, f, indent=2)
    logger.info('Segmented {tiles} tiles in {seconds:.3f}s on {device}'.format(**info))
    return info


def SERVER_test(params, options):
    """Plumbing check (the reference's commented-out SERVER_test, worker.py:300-302):
    writes the params it was called with into the output folder."""
    with open(os.path.join(params['output'], 'test.json'), 'w') as f:
        json.dump({'params': {k: repr(v) for k, v in params.items()},
                   'options': {k: repr(v) for k, v in options.items()}}, f, indent=2)
