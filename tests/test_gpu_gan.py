"""GPU: the progressive WGAN-GP (sequitr/networks/gan.py) on the HIP back end vs the fp64 torch
restatement oracle/torch_gan_ref.py: leaf ops, generator / discriminator forward, the losses with
the gradient penalty (second-order autograd through the hand-written kernels), the per-level
variable lists (including the reference's layer-naming quirk) and one D + G optimiser step."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import torch_gan_ref as ref
from sequitr_amd import functional as F
from sequitr_amd import ops
from sequitr_amd.networks import gan
from tests.util import tiles, rand_weights, assert_bit_exact

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def close(got, want, tol, what=""):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = float(np.max(np.abs(got - want))) / max(float(np.max(np.abs(want))), 1e-30)
    assert err <= tol, "%s: rel err %.3g > %.3g" % (what, err, tol)


def test_leaf_ops_vs_oracle():
    x = tiles(1, 2, 8, 8, 16)
    w = rand_weights(2, (3, 3, 16, 32), 1.0)
    b = rand_weights(3, (32,), 0.1)
    ws = float(np.sqrt(np.float32(2.0 / (9 * 32))))
    assert_bit_exact(ops.conv2d(dev(x), dev(w), dev(b), act="leaky", wscale=ws).cpu().numpy(),
                     co.conv2d(x, w, b, act="leaky", wscale=ws), "weighted_conv2d core")
    assert_bit_exact(ops.broadcast2x2(dev(x), 1.0).cpu().numpy(), co.upsample_nn2x(x), "double_size")
    assert_bit_exact(ops.avgpool2x2(dev(x)).cpu().numpy(), co.avgpool2x2(x), "avg pool")
    for n in (8, 16, 32):                                          # half_size picks rows 0,2,5,7 for 8 -> 4
        img = tiles(4, 1, n, n, 2)
        got = ops.resize_nearest(dev(img), (n // 2, n // 2)).cpu().numpy()
        assert np.array_equal(got, ref.half_size(torch.as_tensor(img)).numpy())
    idx = ops.resize_nearest(dev(np.arange(8, dtype=np.float32).reshape(1, 8, 1, 1)), (4, 1)).cpu().numpy().ravel()
    assert idx.tolist() == [0.0, 2.0, 5.0, 7.0]
    xs = tiles(5, 6, 4, 4, 32)
    want = float(np.sqrt(xs.astype(np.float64).var(0).mean()))
    assert abs(ops.mbstd(dev(xs)).item() - want) <= 1e-6 * want
    a, b2 = tiles(6, 3, 8, 8, 4), tiles(7, 3, 8, 8, 4)
    r = np.array([0.25, 0.5, 1.0], np.float32)
    assert np.allclose(ops.lerp(dev(a), dev(b2), dev(r)).cpu().numpy(), r[:, None, None, None] * a + (1 - r[:, None, None, None]) * b2, atol=1e-6)
    assert np.allclose(ops.lerp(dev(a), dev(b2), 0.3).cpu().numpy(), 0.3 * a + 0.7 * b2, atol=1e-6)
    assert np.allclose(ops.dot_per_sample(dev(a), dev(b2)).cpu().numpy(), (a * b2).reshape(3, -1).sum(1), rtol=1e-5)
    m = ops.wgrad1x1_small(dev(tiles(8, 2, 8, 8, 2)), dev(tiles(9, 2, 8, 8, 16))).cpu().numpy()
    assert np.allclose(m, tiles(8, 2, 8, 8, 2).reshape(-1, 2).T @ tiles(9, 2, 8, 8, 16).reshape(-1, 16), rtol=1e-5, atol=1e-5)


def test_pixelnorm_first_and_second_order():
    x, g, v = tiles(10, 2, 4, 4, 32), tiles(11, 2, 4, 4, 32), tiles(12, 2, 4, 4, 32)
    xt = torch.as_tensor(x, dtype=torch.float64).requires_grad_(True)
    gt = torch.as_tensor(g, dtype=torch.float64).requires_grad_(True)
    dx = torch.autograd.grad(ref.pixel_norm(xt), xt, gt, create_graph=True)[0]
    dx2, dg = torch.autograd.grad(dx, [xt, gt], torch.as_tensor(v, dtype=torch.float64))
    close(ops.pixelnorm_bwd(dev(x), dev(g)).cpu().numpy(), dx.detach().numpy(), 1e-5, "pixelnorm bwd")
    got_dg, got_dx2 = ops.pixelnorm_bwd2(dev(x), dev(g), dev(v))
    close(got_dg.cpu().numpy(), dg.numpy(), 1e-5, "pixelnorm bwd2 dg")
    close(got_dx2.cpu().numpy(), dx2.numpy(), 1e-5, "pixelnorm bwd2 dx")


def test_double_backward_through_conv_stack():
    """grad-of-grad: penalty-style objective through conv(leaky) -> pixel_norm -> avgpool -> conv."""
    x = tiles(13, 2, 8, 8, 2)
    p = {"w0": rand_weights(14, (1, 1, 2, 16), 1.0), "b0": rand_weights(15, (16,), 0.1),
         "w1": rand_weights(16, (3, 3, 16, 16), 1.0), "b1": rand_weights(17, (16,), 0.1)}

    def run(xin, P, conv, pn, pool):
        h = pn(conv(xin, P["w0"], P["b0"], 0.7))
        h = pool(conv(h, P["w1"], P["b1"], 0.2))
        return h

    P = {k: dev(v).requires_grad_(True) for k, v in p.items()}
    xin = dev(x).requires_grad_(True)
    out = run(xin, P, lambda t, w, b, s: F.conv2d(t, w, b, act="leaky", wscale=s), F.pixel_norm, F.avgpool2x2)
    g = torch.autograd.grad(out.sum(), xin, create_graph=True)[0]
    obj = torch.square(torch.sqrt(F.dot_per_sample(g, g)) - 1.0).sum()
    got = torch.autograd.grad(obj, [P[k] for k in sorted(P)])

    R = {k: torch.as_tensor(v, dtype=torch.float64).requires_grad_(True) for k, v in p.items()}
    xr = torch.as_tensor(x, dtype=torch.float64).requires_grad_(True)

    def rconv(t, w, b, s):
        y = torch.nn.functional.conv2d(t.permute(0, 3, 1, 2), (w * s).permute(3, 2, 0, 1), b, padding=w.shape[0] // 2)
        return ref.lrelu(y.permute(0, 2, 3, 1))

    outr = run(xr, R, rconv, ref.pixel_norm, ref.avgpool)
    gr = torch.autograd.grad(outr.sum(), xr, create_graph=True)[0]
    objr = torch.square(torch.sqrt((gr * gr).sum((1, 2, 3))) - 1.0).sum()
    want = torch.autograd.grad(objr, [R[k] for k in sorted(R)])
    close(g.detach().cpu().numpy(), gr.detach().numpy(), 1e-5, "input gradient")
    assert abs(obj.item() - objr.item()) <= 1e-4 * abs(objr.item())
    for k, a, b in zip(sorted(P), got, want):
        close(a.cpu().numpy(), b.numpy(), 2e-4, "d obj / d " + k)


PARAMS = {"num_levels": 3, "batch_size": 4, "repeat_batch": 1, "num_epochs_per_level": 1, "learning_rate": 1e-3,
          "device": "cuda:0", "seed": 3, "num_batches_per_epoch": 2}


def make_gan(**kw):
    g = gan.GenerativeAdverserialNetwork(dict(PARAMS, **kw), mode=None)
    g.build()
    return g


def test_build_creates_reference_variable_names_and_var_lists():
    g = make_gan()
    assert g.filters == [32, 16, 8] and g.get_size(2) == (16, 16)
    names = list(g.store.vars)
    for n in ("GAN/generator/latent/dense1/kernel", "GAN/generator/latent/conv/filter", "GAN/generator/layer_0/conv1/filter",
              "GAN/generator/layer_1/conv2/bias", "GAN/generator/to_image/to_image2/filter",
              "GAN/discriminator/from_image/from_image0/filter", "GAN/discriminator/from_image/from_image2/bias",
              "GAN/discriminator/layer_1/conv1/filter", "GAN/discriminator/layer_2/conv2/filter",
              "GAN/discriminator/output/conv/filter", "GAN/discriminator/output/dense/kernel",
              "GAN/discriminator/output/logits/bias"):
        assert n in names, n
    assert g.store.vars["GAN/generator/latent/dense1/kernel"].shape == (512, 16 * 32)
    assert g.store.vars["GAN/discriminator/output/dense/kernel"].shape == (16 * 33, 32)
    assert g.store.vars["GAN/discriminator/from_image/from_image2/filter"].shape == (1, 1, 2, 8)
    assert not any("discriminator/layer_0" in n for n in names)            # blocks are named layer_L .. layer_1
    d2, g2 = g.get_training_variables(2)
    dn, gn = [n for n, _ in d2], [n for n, _ in g2]
    assert any("discriminator/layer_1/" in n for n in dn) and any("from_image2" in n for n in dn)
    assert not any("discriminator/layer_2/" in n for n in dn)              # the newest block is NOT trained (a25)
    assert any("generator/layer_1/" in n for n in gn) and any("to_image2" in n for n in gn) and any("latent" in n for n in gn)
    assert not any("to_image1" in n for n in gn)


@pytest.mark.parametrize("level", [0, 1, 2])
def test_generator_discriminator_forward_vs_fp64(level):
    g = make_gan()
    W = ref.to_torch(g.store.state_dict(), requires_grad=False)
    f = g.filters[:level + 1]
    z = np.random.default_rng(0).standard_normal((4, 1, 1, 512)).astype(np.float32)
    with torch.no_grad():
        outs, last = g.generator(dev(z), f)
    routs, rlast = ref.generator(torch.as_tensor(z, dtype=torch.float64), W, f)
    assert len(outs) == level + 1 and tuple(last.shape) == (4,) + g.get_size(level) + (2,)
    for a, b in zip(outs, routs):
        close(a.cpu().numpy(), b.numpy(), 2e-5, "generator image")
    x = np.random.default_rng(1).standard_normal((4,) + g.get_size(level) + (2,)).astype(np.float32)
    with torch.no_grad():
        layers, logits = g.discriminator(dev(x), f[::-1])
    close(logits.cpu().numpy(), ref.discriminator(torch.as_tensor(x, dtype=torch.float64), W, f[::-1]).numpy(), 2e-5, "D logits")
    assert len(layers) == level + 1


@pytest.mark.parametrize("level,alpha", [(0, 1.0), (2, 0.4), (2, 1.0)])
def test_losses_and_gradients_vs_fp64(level, alpha):
    g = make_gan()
    g.set_level(level)
    rng = np.random.default_rng(2)
    z = rng.standard_normal((4, 1, 1, 512)).astype(np.float32)
    x = rng.standard_normal((4,) + g.get_size(level) + (2,)).astype(np.float32)
    r = rng.random(4).astype(np.float32)
    _, d_loss, g_loss = g._build_network(dev(x), dev(z), alpha, r=dev(r))
    d_vars, g_vars = g.get_training_variables(level)
    dg = torch.autograd.grad(d_loss, [v for _, v in d_vars], retain_graph=True, allow_unused=True)
    gg = torch.autograd.grad(g_loss, [v for _, v in g_vars], allow_unused=True)

    W = ref.to_torch(g.store.state_dict())
    _, rd, rg = ref.losses(torch.as_tensor(x, dtype=torch.float64), torch.as_tensor(z, dtype=torch.float64), alpha,
                           torch.as_tensor(r, dtype=torch.float64), W, g.filters, level)
    assert abs(d_loss.item() - rd.item()) <= 2e-4 * max(1.0, abs(rd.item()))
    assert abs(g_loss.item() - rg.item()) <= 2e-4 * max(1.0, abs(rg.item()))
    rdg = torch.autograd.grad(rd, [W[n] for n, _ in d_vars], retain_graph=True, allow_unused=True)
    rgg = torch.autograd.grad(rg, [W[n] for n, _ in g_vars], allow_unused=True)
    for (n, _), a, b in zip(d_vars, dg, rdg):
        close(a.cpu().numpy(), b.numpy(), 2e-3, "d_loss grad " + n)
    for (n, _), a, b in zip(g_vars, gg, rgg):
        close(a.cpu().numpy(), b.numpy(), 2e-3, "g_loss grad " + n)


def test_one_solver_step_and_training_loop(tmp_path):
    g = make_gan(output=str(tmp_path / "gan_out"))
    g.set_level(1)
    rng = np.random.default_rng(4)
    z = dev(rng.standard_normal((4, 1, 1, 512)).astype(np.float32))
    x = dev(rng.standard_normal((4, 8, 8, 2)).astype(np.float32))
    before = g.store.state_dict()
    d_vars, g_vars = g.get_training_variables(1)
    g.d_solver(x, z, 0.5)
    mid = g.store.state_dict()
    g.g_solver(x, z, 0.5)
    after = g.store.state_dict()
    dn, gn = {n for n, _ in d_vars}, {n for n, _ in g_vars}
    for k in before:
        if k in dn:
            delta = np.abs(mid[k] - before[k])
            assert delta.max() <= 1.01e-3 and delta.max() > 0, k         # |step| <= lr at t = 1 (beta1 = 0)
        else:
            assert np.array_equal(mid[k], before[k]), k
        if k in gn:
            assert np.abs(after[k] - mid[k]).max() > 0, k
        else:
            assert np.array_equal(after[k], mid[k]), k
    assert g.global_step == 1 and g.d_opt.t == 1 and g.g_opt.t == 1
    # the reference's level / phase / repeat loop (gan.py:823-867), 2 steps per phase
    g.train(max_steps_per_phase=2)
    assert g.global_step == 1 + 3 * 2 * 2 and all(np.isfinite(v) for v in g.last_losses)
    import os
    assert sorted(os.listdir(str(tmp_path / "gan_out"))) == ["model_(16x16).npz", "model_(4x4).npz", "model_(8x8).npz"]
    img = g.predict(latent=np.zeros((2, 1, 1, 512), np.float32))
    assert tuple(img.shape) == (2, 16, 16, 2)
    h = gan.GenerativeAdverserialNetwork(dict(PARAMS), mode=None)
    h.load_checkpoint(str(tmp_path / "gan_out" / "model_(16x16).npz"))
    assert torch.equal(h.predict(latent=np.zeros((2, 1, 1, 512), np.float32)), img)


def test_configuration_defaults():
    c = gan.GAN2DConfiguration()
    p = c.to_params()
    assert p["name"] == "GAN_competition" and p["batch_size"] == 32 and p["repeat_batch"] == 4
    assert p["num_levels"] == 7 and p["start_size"] == (4, 4) and p["learning_rate"] == 1e-3
    from sequitr_amd import utils
    assert utils.filter_doubling(8, 7, 512, True) == [512, 256, 128, 64, 32, 16, 8]


def test_restore_and_convert_checkpoint_to_model(tmp_path):
    """gan.py:811-816 (restore) and :874-903 (convert_checkpoint_to_model): resume from the last finished
    level; export the generator as an inference model."""
    import json
    import os
    out = str(tmp_path / "gan_out")
    g = make_gan(output=out)
    g.train(max_steps_per_phase=1)                              # 3 levels -> 3 checkpoints
    ref_img = g.predict(latent=np.ones((1, 1, 1, 512), np.float32))
    os.remove(os.path.join(out, "model_(16x16).npz"))           # pretend the last level never finished
    h = make_gan(output=out)
    h.restore = True
    assert h.latest_checkpoint()[0] == 1
    steps_before = h.global_step
    h.train(max_steps_per_phase=1)                              # resumes at level 2 only: 2 phases x 1 step
    assert h.global_step - steps_before == 2 and os.path.exists(os.path.join(out, "model_(16x16).npz"))
    export = h.convert_checkpoint_to_model()
    meta = json.load(open(os.path.join(export, "model.json")))
    assert meta["outputs"]["Gz"] == [None, 16, 16, 2] and meta["level"] == 2 and "alpha" in meta["inputs"]
    with np.load(os.path.join(export, "weights.npz")) as z:
        keys = list(z.files)
        assert keys and all(k.startswith("GAN/generator/") for k in keys)
        k = gan.GenerativeAdverserialNetwork(dict(PARAMS), mode=None)
        k.build()
        k.store.load_state_dict({n: z[n] for n in keys})
    assert torch.equal(k.predict(latent=np.ones((1, 1, 1, 512), np.float32)),
                       h.predict(latent=np.ones((1, 1, 1, 512), np.float32)))
    assert tuple(ref_img.shape) == (1, 16, 16, 2)


def test_cli_train_then_predict_exports_tiffs(tmp_path):
    """python -m sequitr_amd.networks.gan --train ... / predict (gan.py:1040-1125): run folder numbering,
    export/ model, export_<i>.tif with the reference's RGB mapping."""
    import os
    work = str(tmp_path)
    os.makedirs(os.path.join(work, "GAN3"))
    export = gan.main(["--workdir", work, "--train", "--num_levels", "2", "--num_epochs", "1", "--batch_size", "4"])
    assert export == os.path.join(work, "GAN4", "export") and os.path.exists(os.path.join(export, "weights.npz"))
    out = gan.main(["--workdir", os.path.join(work, "GAN4"), "--num_levels", "2", "--batch_size", "4", "--samples", "3"])
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(out, "export_2.tif")))
    assert img.shape == (8, 8, 3) and img.dtype == np.uint8 and np.array_equal(img[..., 0], img[..., 2])
    x = np.array([[[[-3.0, 0.0]]], [[[3.0, 9.0]]]])
    assert gan.to_rgb(x).tolist() == [[[[127, 0, 127]]], [[[255, 255, 255]]]]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_graphed_solver_steps_equal_eager_steps(dtype):
    """params['graph']: the solver steps replayed as hipGraphs (gradients | all-reduce | Adam) leave bit-identical
    weights to the eager steps (same kernels, same mixing draws, device-side Adam counter); the fade-in alpha is a
    device tensor, so one pair of graphs serves every alpha."""
    def run(graph):
        g = make_gan(graph=graph, dtype=dtype)
        g.set_level(1)
        rng = np.random.default_rng(12)
        for it, alpha in enumerate((0.25, 0.5, 1.0, 1.0)):
            z = dev(rng.standard_normal((4, 1, 1, 512)).astype(np.float32))
            x = dev(rng.standard_normal((4, 8, 8, 2)).astype(np.float32))
            g.d_solver(x, z, alpha)
            g.g_solver(x, z, alpha)
        g.d_solver(x, z, 0.5)
        g.d_solver(x, z, 1.0)
        return g
    a, b = run(False), run(True)
    assert len(b._graphs) == 2 and all(isinstance(v, tuple) for v in b._graphs.values()) and not a._graphs
    wa, wb = a.store.state_dict(), b.store.state_dict()
    for k in wa:
        assert np.array_equal(wa[k], wb[k]), k
    assert a.global_step == b.global_step == 4 and a.d_opt.t == b.d_opt.t == 6 and a.g_opt.t == b.g_opt.t == 4
    assert a.last_losses == b.last_losses
    assert int(b.d_opt.state[0].item()) == 6


def test_training_loop_with_graph_replay_across_levels(tmp_path):
    """train() with params['graph']: every level captures its own set of iteration graphs (generator forward + discriminator
    gradients, Adam D, generator gradients, Adam G; 4x4 mosaic level included); the fade-in weight rides in a device tensor,
    so fade and stabilisation phases replay the same graphs; checkpoints and global_step as in the eager loop."""
    g = make_gan(output=str(tmp_path / "o"), graph=True, dtype="bf16")
    g.train(max_steps_per_phase=3)
    assert g.global_step == 3 * 2 * 2 and all(np.isfinite(v) for v in g.last_losses)     # 2 batches per epoch cap the phase
    captured = [k for k, v in g._graphs.items() if isinstance(v, tuple)]
    assert sorted((k[0], k[1]) for k in captured) == [("it", 0), ("it", 1), ("it", 2)]
    assert all(len(g._graphs[k]) == 14 for k in captured)                  # four graphs + their static tensors and tables
    import os
    assert sorted(os.listdir(str(tmp_path / "o"))) == ["model_(16x16).npz", "model_(4x4).npz", "model_(8x8).npz"]
    w = g.store.state_dict()
    assert all(np.isfinite(v).all() for v in w.values())


@pytest.mark.parametrize("level,alpha", [(0, 1.0), (2, 0.6)])
def test_stacked_discriminator_pass_equals_separate_passes(level, alpha):
    """params['batch_d']: D(Gz) and D(X) as one pass over the stacked minibatches, each with its own minibatch
    statistic: the same per-sample arithmetic (losses to f32 rounding), gradients equal up to summation order."""
    res = {}
    for bd in (True, False):
        g = make_gan(batch_d=bd)
        g.set_level(level)
        rng = np.random.default_rng(2)
        z = rng.standard_normal((4, 1, 1, 512)).astype(np.float32)
        x = rng.standard_normal((4,) + g.get_size(level) + (2,)).astype(np.float32)
        r = rng.random(4).astype(np.float32)
        _, d_loss, g_loss = g._build_network(dev(x), dev(z), alpha, r=dev(r))
        d_vars, g_vars = g.get_training_variables(level)
        dg = torch.autograd.grad(d_loss, [v for _, v in d_vars], retain_graph=True, allow_unused=True)
        gg = torch.autograd.grad(g_loss, [v for _, v in g_vars], allow_unused=True)
        res[bd] = (d_loss.item(), g_loss.item(), [t.cpu().numpy() for t in dg], [t.cpu().numpy() for t in gg])
    a, b = res[True], res[False]
    assert abs(a[0] - b[0]) <= 1e-5 * max(1.0, abs(b[0])) and abs(a[1] - b[1]) <= 1e-5 * max(1.0, abs(b[1]))
    for u, v in zip(a[2] + a[3], b[2] + b[3]):
        close(u, v, 5e-4, "gradient, stacked vs separate")     # f32 sums in another order (vs fp64: 2e-3 for either)
    # the minibatch statistic is per group, not over the stacked batch
    feat = torch.randn(8, 4, 4, 16, device="cuda:0")
    m = gan.minibatch_stdev(feat, groups=2)
    assert torch.allclose(m[:4], gan.minibatch_stdev(feat[:4])) and torch.allclose(m[4:], gan.minibatch_stdev(feat[4:]))
    assert not torch.allclose(m[:4], gan.minibatch_stdev(feat)[:4])


# ---- BASELINE configs[4] at FULL size: level 6 = 256x256x2, batch 32, filters [512 ... 8] ----------
FULL = {"num_levels": 7, "batch_size": 32, "repeat_batch": 1, "learning_rate": 1e-3, "device": "cuda:0", "seed": 0}


def _full_inputs(n=32):
    rng = np.random.default_rng(3)
    return (dev(rng.standard_normal((n, 256, 256, 2)).astype(np.float32)),
            dev(rng.standard_normal((n, 1, 1, 512)).astype(np.float32)))


def test_config5_level6_forward_vs_fp64_batch2():
    """generator / discriminator at level 6 with the reference's filter schedule vs oracle/torch_gan_ref.py (fp64)."""
    g = gan.GenerativeAdverserialNetwork(dict(FULL), mode=None)
    g.build()
    assert g.filters == [512, 256, 128, 64, 32, 16, 8] and g.get_size(6) == (256, 256)
    W = ref.to_torch(g.store.state_dict(), requires_grad=False)
    rng = np.random.default_rng(5)
    z = rng.standard_normal((2, 1, 1, 512)).astype(np.float32)
    x = rng.standard_normal((2, 256, 256, 2)).astype(np.float32)
    with torch.no_grad():
        outs, last = g.generator(dev(z), g.filters)
        _, logits = g.discriminator(dev(x), g.filters[::-1])
    routs, _ = ref.generator(torch.as_tensor(z, dtype=torch.float64), W, g.filters)
    assert tuple(last.shape) == (2, 256, 256, 2) and len(outs) == 7
    for a, b in zip(outs, routs):
        close(a.cpu().numpy(), b.numpy(), 2e-4, "generator image, level-6 schedule")   # f32 through 14 conv layers, K up to 4608: 8.5e-5 measured
    close(logits.cpu().numpy(), ref.discriminator(torch.as_tensor(x, dtype=torch.float64), W, g.filters[::-1]).numpy(),
          2e-4, "D logits, level 6")


def test_config5_level6_batch32_bf16_replay_equals_eager():
    """one d_solver + g_solver at config 5's full size in its dtype (bf16-multiply convolutions): finite losses, and
    three iterations (eager warm-up, capture, replay) leave bit-identical weights with and without hipGraph replay."""
    X, Z = _full_inputs()

    def run(graph):
        g = gan.GenerativeAdverserialNetwork(dict(FULL, dtype="bf16", graph=graph), mode=None)
        g.build()
        g.set_level(6)
        for _ in range(3):
            g.d_solver(X, Z, 1.0)
            g.g_solver(X, Z, 1.0)
        assert all(np.isfinite(v) for v in g.last_losses)
        return g
    a, b = run(False), run(True)
    assert len(b._graphs) == 2 and all(isinstance(v, tuple) for v in b._graphs.values())
    assert a.last_losses == b.last_losses
    wa, wb = a.store.state_dict(), b.store.state_dict()
    changed = 0
    for k in wa:
        assert np.array_equal(wa[k], wb[k]), k
        assert np.isfinite(wa[k]).all(), k
    w0 = gan.GenerativeAdverserialNetwork(dict(FULL), mode=None)
    w0.build()
    w0 = w0.store.state_dict()
    changed = sum(1 for k in wa if not np.array_equal(wa[k], w0[k]))
    assert changed >= 40, changed                                 # every trained D and G variable moved


def test_config5_level6_batch32_stacked_d_equals_separate_passes():
    """params['batch_d'] at full size (f32: the exact-f32 kernels make the comparison sharp)."""
    X, Z = _full_inputs()
    r = dev(np.random.default_rng(6).random(32).astype(np.float32))
    res = {}
    for bd in (True, False):
        g = gan.GenerativeAdverserialNetwork(dict(FULL, batch_d=bd), mode=None)
        g.build()
        g.set_level(6)
        _, d_loss, g_loss = g._build_network(X, Z, 1.0, r=r, need_g_graph=False)
        d_vars, _ = g.get_training_variables(6)
        dg = torch.autograd.grad(d_loss, [v for _, v in d_vars], allow_unused=True)
        res[bd] = (d_loss.item(), g_loss.item(), [None if t is None else t.cpu().numpy() for t in dg])
        res["names"] = [(n, None) for n, _ in d_vars]
        del g, dg, d_loss, g_loss
        torch.cuda.empty_cache()
    a, b = res[True], res[False]
    assert abs(a[0] - b[0]) <= 1e-5 * max(1.0, abs(b[0])) and abs(a[1] - b[1]) <= 1e-5 * max(1.0, abs(b[1]))
    rows = []
    for (name, _), u, v in zip(res["names"], a[2], b[2]):
        assert (u is None) == (v is None), name
        if u is not None:
            rows.append((name, float(np.abs(u - v).max()), float(np.abs(v).max())))
    # 2e-7 absolute: d(d_loss)/d(logits bias) = mean(-1 + 1 + 0.002 Dx) cancels to ~1e-5, so its f32 rounding (1e-7) is
    # 1 % of it; every other gradient agrees to ~1e-6 of its largest element (measured)
    worst = max(rows, key=lambda t: (t[1] - 2e-7) / max(t[2], 1e-30))
    assert worst[1] <= 5e-4 * worst[2] + 2e-7, "stacked vs separate, level 6: %s; all: %s" % (worst, rows)


def test_nested_precision_blocks_do_not_reuse_stale_filter_packs():
    """under an outer `with net.precision():` the inner solver blocks used to leave the packed bf16 filters
    cached across the Adam update (eager: stale weights in the next step; graphed: no pack kernel captured).  The cache
    is now invalidated by the weight update itself: nested and un-nested runs give the same bits, eager and graphed."""
    rng = np.random.default_rng(12)
    z = dev(rng.standard_normal((4, 1, 1, 512)).astype(np.float32))
    x = dev(rng.standard_normal((4, 8, 8, 2)).astype(np.float32))

    def run(graph, nested):
        g = make_gan(graph=graph, dtype="bf16")
        g.set_level(1)
        import contextlib
        with (g.precision() if nested else contextlib.nullcontext()):
            for _ in range(4):
                g.d_solver(x, z, 1.0)
                g.g_solver(x, z, 1.0)
            assert not nested or ops.MIXED
            assert not ops._PACKS                                  # nothing survives a solver step
        return g.store.state_dict()
    want = run(False, False)
    for graph, nested in ((False, True), (True, False), (True, True)):
        got = run(graph, nested)
        for k in want:
            assert np.array_equal(want[k], got[k]), (graph, nested, k)


@pytest.mark.parametrize("groups", [1, 2])
def test_mbstd_map_first_and_second_order_vs_fp64(groups):
    """sq_mbstd_map_{fwd,bwd,bwd2}_f32 (the discriminator's minibatch-stdev feature, gan.py:204-212) against torch
    autograd in fp64 of the reference expression: value, gradient, and the gradient of a function of the gradient
    (what the WGAN-GP penalty needs), per stacked minibatch group."""
    n, C = 6, 32
    x = tiles(21, groups * n, 4, 4, C)
    cot = tiles(22, groups * n, 4, 4, 1)
    cot2 = tiles(23, groups * n, 4, 4, C)

    def ref_map(xt):
        outs = []
        for g in range(groups):
            xg = xt[g * n:(g + 1) * n]
            s = torch.sqrt(xg.var(dim=0, unbiased=False).mean())
            outs.append(torch.ones((n, 4, 4, 1), dtype=xt.dtype) * s)
        return torch.cat(outs, 0)
    xr = torch.as_tensor(x, dtype=torch.float64).requires_grad_(True)
    cr = torch.as_tensor(cot, dtype=torch.float64).requires_grad_(True)
    yr = ref_map(xr)
    g1 = torch.autograd.grad(yr, xr, cr, create_graph=True)[0]
    obj = (g1 * torch.as_tensor(cot2, dtype=torch.float64)).sum()
    gx2, gc = torch.autograd.grad(obj, [xr, cr])

    xd, cd = dev(x).requires_grad_(True), dev(cot).requires_grad_(True)
    y = gan.minibatch_stdev(xd, groups)
    assert tuple(y.shape) == (groups * n, 4, 4, 1)
    close(y.detach().cpu().numpy(), yr.detach().numpy(), 1e-6, "mbstd map")
    d1 = torch.autograd.grad(y, xd, cd, create_graph=True)[0]
    close(d1.detach().cpu().numpy(), g1.detach().numpy(), 1e-5, "mbstd map gradient")
    o = (d1 * dev(cot2)).sum()
    dx2, dc = torch.autograd.grad(o, [xd, cd])
    close(dx2.cpu().numpy(), gx2.numpy(), 1e-4, "mbstd map second order, d/dx")
    close(dc.cpu().numpy(), gc.numpy(), 1e-4, "mbstd map second order, d/d(dy)")


def test_fused_wgan_losses_vs_the_reference_expression():
    """sq_wgan_losses_{fwd,bwd}_f32 against gan.py:715-729 written out in torch fp64 (one-sided penalty: samples with
    |grad| < 1 contribute nothing, also to the gradient)."""
    rng = np.random.default_rng(30)
    N = 32
    dz, dx = rng.standard_normal(N).astype(np.float32), rng.standard_normal(N).astype(np.float32) * 3
    gn2 = (rng.random(N).astype(np.float32) * 3.0) ** 2            # norms on both sides of 1
    tz, tx, tg = [torch.as_tensor(a, dtype=torch.float64).requires_grad_(True) for a in (dz, dx, gn2)]
    pen = 10.0 * torch.square(torch.clamp(torch.sqrt(tg) - 1.0, min=0.0))
    rd, rg = torch.mean(-tx + tz + pen + 0.001 * torch.square(tx)), torch.mean(-tz)
    gz, gxx, ggn = torch.autograd.grad(2.0 * rd + 3.0 * rg, [tz, tx, tg])
    z, xx, g2 = [dev(a).requires_grad_(True) for a in (dz, dx, gn2)]
    d_loss, g_loss = F.wgan_losses(z, xx, g2)
    assert abs(d_loss.item() - rd.item()) <= 1e-5 * abs(rd.item()) and abs(g_loss.item() - rg.item()) <= 1e-6
    a, b, c = torch.autograd.grad(2.0 * d_loss + 3.0 * g_loss, [z, xx, g2])
    close(a.cpu().numpy(), gz.numpy(), 1e-5, "dDz")
    close(b.cpu().numpy(), gxx.numpy(), 1e-5, "dDx")
    close(c.cpu().numpy(), ggn.numpy(), 1e-5, "d gn2")
    assert (c.cpu().numpy()[np.sqrt(gn2) < 1.0] == 0).all()
    _, only_g = F.wgan_losses(z)
    assert abs(only_g.item() - rg.item()) <= 1e-6
    (a2,) = torch.autograd.grad(only_g, [z])
    assert np.allclose(a2.cpu().numpy(), -1.0 / N)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_act_gate_fusion_gives_the_same_bits(dtype):
    """functional.ActGate: in first-order backward passes the leaky-ReLU backward of a convolution rides in the kernel
    that produces the gradient (pixel_norm backward in the generator, the average pool's up-sampling in the
    discriminator) instead of a separate act_bwd pass.  Same multiply on the same values: weights after D and G
    steps are bit-identical with the fusion on (default for the reference's networks) and off."""
    rng = np.random.default_rng(12)
    z = dev(rng.standard_normal((4, 1, 1, 512)).astype(np.float32))
    x = dev(rng.standard_normal((4, 16, 16, 2)).astype(np.float32))

    def run(fuse):
        g = make_gan(dtype=dtype)
        g._default_g = fuse                                     # the switch the solver steps read
        g.set_level(2)
        for alpha in (0.5, 1.0):
            g.d_solver(x, z, alpha)
            g.g_solver(x, z, alpha)
        return g.store.state_dict(), g.last_losses
    (wa, la), (wb, lb) = run(True), run(False)
    assert la == lb
    for k in wa:
        assert np.array_equal(wa[k], wb[k]), k
    assert not F.FUSE_ACT_GATES


@pytest.mark.parametrize("dtype,graph", [("f32", False), ("bf16", False), ("bf16", True)])
def test_iteration_shares_the_generator_pass_and_equals_the_two_solver_calls(dtype, graph):
    """GenerativeAdverserialNetwork.iteration(X, Z, alpha): d_solver + g_solver on one feed (gan.py:848-851) with ONE generator
    forward pass -- the discriminator step does not move generator weights, so the pass the generator step repeats is the
    one already evaluated.  Weights and losses after three iterations (fade and stabilisation alphas) are bit-identical to
    the separate calls, eager and as replayed hipGraphs (call 1 eager, call 2 captures, call 3 replays)."""
    rng = np.random.default_rng(21)
    feeds = [(dev(rng.standard_normal((4, 16, 16, 2)).astype(np.float32)), dev(rng.standard_normal((4, 1, 1, 512)).astype(np.float32)))
             for _ in range(4)]
    alphas = (0.5, 1.0, 1.0, 0.75)

    def run(shared):
        g = make_gan(dtype=dtype, graph=graph)
        g.set_level(2)
        out = []
        for (x, z), a in zip(feeds, alphas):
            if shared:
                d, gl = g.iteration(x, z, a)
            else:
                d, gl = g.d_solver(x, z, a), g.g_solver(x, z, a)
            out.append((float(d), float(gl)))
        return g.store.state_dict(), out, g.global_step
    (wa, la, sa), (wb, lb, sb) = run(True), run(False)
    assert la == lb and sa == sb == 4
    for k in wa:
        assert np.array_equal(wa[k], wb[k]), k


@pytest.mark.parametrize("dtype,level", [("f32", 0), ("f32", 2), ("bf16", 2)])
def test_act_gates_with_an_active_penalty_give_the_ungated_gradients(dtype, level):
    """Round 4 regression: with the WGAN-GP penalty ACTIVE (|dD(mix)/dmix| > 1) the second-order pass sends from_image's
    activation a second gradient (pixel norm's dL/dx beside its first-order backward); the pixel-norm ActGate must then
    leave act' to the conv.  A freshly initialised deep discriminator has gradient norms below 1 (penalty exactly zero),
    which is why the bit-identity test above never saw it: here the last layer is scaled until the penalty dominates
    d_loss."""
    rng = np.random.default_rng(5)
    z = dev(rng.standard_normal((4, 1, 1, 512)).astype(np.float32))
    x = dev(rng.standard_normal((4,) + (4 * 2 ** level,) * 2 + (2,)).astype(np.float32))
    r = dev(rng.random(4).astype(np.float32))

    def run(fuse):
        g = make_gan(dtype=dtype)
        g.set_level(level)
        with torch.no_grad():
            g.store.vars["GAN/discriminator/output/logits/kernel"].mul_(300.0)
        ops.invalidate_packs()
        with g.precision(), F.fuse_act_gates(fuse):
            g._pack_filters()
            names, grads, (d_loss, g_loss) = g._d_grads(x, z, 0.6, r)
        return [n for n, _ in names], [None if t is None else t.clone() for t in grads], d_loss.item(), g_loss.item()
    na, ga, da, gla = run(True)
    nb_, gb, db_, glb = run(False)
    assert da == db_ and da > 10.0 * abs(gla) + 10.0               # the penalty is what d_loss is made of
    for n, u, v in zip(na, ga, gb):
        assert (u is None) == (v is None), n
        if u is not None:
            assert torch.equal(u, v), (n, float((u - v).abs().max()), float(v.abs().max()))


@pytest.mark.parametrize("M,K,N", [(64, 8208, 512), (32, 512, 8192), (5, 300, 260), (128, 1024, 64)])
def test_dense_weight_gradient_kernel(M, K, N):
    """sq_dense_wgrad_f32: dW = scale * x^T dY and db = column sums of dY for a dense layer's few rows (the discriminator's
    8208 -> 512 layer, the generator's 512 -> 8192 latent layer), exact-f32 fmaf chains in row order; also through
    conv2d_wgrad's dispatch for the (1,1,M,K) form F.dense runs."""
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    dy = rng.standard_normal((M, N)).astype(np.float32)
    dw, db = ops.dense_wgrad(dev(x), dev(dy), want_bias=True, dw_scale=0.5)
    rw = 0.5 * (x.astype(np.float64).T @ dy.astype(np.float64))
    assert np.abs(dw.cpu().numpy() - rw).max() <= 2e-6 * np.abs(rw).max() * np.sqrt(M)
    assert np.allclose(db.cpu().numpy(), dy.astype(np.float64).sum(0), rtol=1e-5, atol=1e-5)
    if K * N >= (1 << 16):
        dw4, db4 = ops.conv2d_wgrad(dev(x.reshape(1, 1, M, K)), dev(dy.reshape(1, 1, M, N)), 1, want_bias=True, dw_scale=0.5)
        assert tuple(dw4.shape) == (1, 1, K, N) and torch.equal(dw4.view(K, N), dw) and torch.equal(db4, db)
