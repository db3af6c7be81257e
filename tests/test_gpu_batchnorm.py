"""GPU parity of the optional batch normalisation of conv_layer (SURVEY.md A.1 `batch_norm`; the
reference leaves conv_layer abstract, sequitr/networks/unet.py:326-328 -> parity unpinned, the
contract is oracle/sq_oracle.c's BN restatement + the fp64 torch graph for gradients)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import torch_ref as tr
from oracle import unet_oracle
from sequitr_amd import ops
from sequitr_amd.networks.unet import UNet2D, init_unet_weights
from sequitr_amd.train import UNetTrainer
from tests.util import tiles, assert_bit_exact

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


@pytest.mark.parametrize("shape", [(2, 16, 16, 16), (3, 9, 7, 64), (1, 5, 5, 256), (4, 33, 17, 8)])
def test_bn_stats_fold_apply_vs_oracle(shape):
    rng = np.random.default_rng(sum(shape))
    x = (rng.standard_normal(shape) * 1.7 + 0.3).astype(np.float32)
    C = shape[-1]
    gamma, beta = rng.standard_normal(C).astype(np.float32), rng.standard_normal(C).astype(np.float32)
    mean, var = ops.bn_stats(dev(x))
    rmean, rvar = co.bn_stats(x)
    # fp64 sums in a different (fixed) order: equal to the last float bit except on rounding ties
    assert np.allclose(mean.cpu().numpy(), rmean, rtol=2e-7, atol=1e-8)
    assert np.allclose(var.cpu().numpy(), rvar, rtol=1e-6, atol=1e-9)
    scale, shift = ops.bn_fold(dev(gamma), dev(beta), dev(rmean), dev(rvar), 1e-3)
    rscale, rshift = co.bn_fold(gamma, beta, rmean, rvar, 1e-3)
    assert_bit_exact(scale.cpu().numpy(), rscale, "scale"), assert_bit_exact(shift.cpu().numpy(), rshift, "shift")
    for act in (None, "relu", "leaky"):
        y = ops.bn_apply(dev(x), dev(rscale), dev(rshift), act)
        assert_bit_exact(y.cpu().numpy(), co.bn_apply(x, rscale, rshift, act), "apply %s" % act)


def test_bn_moving_update_matches_tf_rule():
    rng = np.random.default_rng(0)
    mm, mv = rng.standard_normal(16).astype(np.float32), (1 + rng.random(16)).astype(np.float32)
    m, v = rng.standard_normal(16).astype(np.float32), rng.random(16).astype(np.float32)
    dm, dv = dev(mm), dev(mv)
    ops.bn_update_moving_(dm, dv, dev(m), dev(v), npix=100, momentum=0.99)
    assert np.allclose(dm.cpu().numpy(), mm * 0.99 + m * 0.01, rtol=1e-6, atol=1e-7)
    assert np.allclose(dv.cpu().numpy(), mv * 0.99 + v * (100 / 99.0) * 0.01, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("act", [None, "relu"])
def test_bn_backward_vs_fp64_autograd(act):
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((2, 12, 10, 32)) * 1.5 + 0.2).astype(np.float32)
    dy = rng.standard_normal(x.shape).astype(np.float32)
    gamma, beta = (1 + 0.3 * rng.standard_normal(32)).astype(np.float32), rng.standard_normal(32).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    gt = torch.tensor(gamma, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(beta, dtype=torch.float64, requires_grad=True)
    mu = xt.mean((0, 1, 2))
    var = ((xt - mu) ** 2).mean((0, 1, 2))
    yt = gt * (xt - mu) / torch.sqrt(var + 1e-3) + bt
    if act:
        yt = torch.relu(yt)
    (yt * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    mean, v = ops.bn_stats(dev(x))
    scale, shift = ops.bn_fold(dev(gamma), dev(beta), mean, v, 1e-3)
    y = ops.bn_apply(dev(x), scale, shift, act)
    dx, dgamma, dbeta = ops.bn_bwd(dev(x), dev(dy), y, act, mean, v, dev(gamma), 1e-3)
    for got, ref, name in ((dx, xt.grad, "dx"), (dgamma, gt.grad, "dgamma"), (dbeta, bt.grad, "dbeta")):
        ref = ref.numpy()
        assert np.max(np.abs(got.cpu().numpy() - ref)) <= 2e-5 * np.max(np.abs(ref)) + 1e-6, name


def test_unet_inference_with_batchnorm_bit_exact_vs_oracle():
    params = {"shape": (32, 32), "filters": (16, 32, 64), "batch_norm": True}
    w = init_unet_weights(params, 2)
    rng = np.random.default_rng(3)
    for k in [k for k in w if k.endswith("gamma")]:                 # non-trivial BN state
        n = w[k].shape[0]
        w[k] = (1 + 0.2 * rng.standard_normal(n)).astype(np.float32)
        w[k[:-5] + "beta"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
        w[k[:-5] + "moving_mean"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
        w[k[:-5] + "moving_variance"] = (0.5 + rng.random(n)).astype(np.float32)
    net = UNet2D(dict(params, device="cuda:0"), "infer")
    net.load_state_dict(w)
    x = tiles(5, 2, 32, 32)
    mask = net.predict(x)
    ref_logits, ref_net = unet_oracle.unet_forward(x, w, params, return_net=True)
    for i, (a, b) in enumerate(zip(net._net, ref_net)):
        assert_bit_exact(a.cpu().numpy(), b, "layer %d" % i)
    assert_bit_exact(mask.cpu().numpy(), unet_oracle.predict_mask(ref_logits), "mask")


def test_unet_training_with_batchnorm_vs_fp64():
    params = {"shape": (32, 32), "filters": (16, 32, 64), "batch_norm": True, "dropout": 0.0,
              "device": "cuda:0", "seed": 4}
    rng = np.random.default_rng(6)
    x = tiles(6, 3, 32, 32)
    lab = rng.random((3, 32, 32)) < 0.4
    onehot = np.stack([~lab, lab], -1).astype(np.uint8)
    wmap = (1 + 3 * rng.random((3, 32, 32, 1))).astype(np.float32)
    t = UNetTrainer(params, learning_rate=0.01)
    w0 = {k: v for k, v in t.state_dict().items()}
    assert any(k.endswith("gamma") for k in t.pbucket.names)
    loss = t.forward_backward(dev(x), dev(onehot), dev(wmap))
    rloss, rgrads, _ = tr.unet_loss_and_grads(x, onehot, wmap, {k: w0[k] for k in t.pbucket.names}, params)
    assert abs(loss.item() - rloss) <= 2e-5 * abs(rloss)
    g = t.grads()
    for k in rgrads:
        scale = np.max(np.abs(rgrads[k])) + 1e-12
        # biases in front of a BN layer have an exactly-zero gradient (the batch mean removes them)
        assert np.max(np.abs(g[k] - rgrads[k])) <= 2e-3 * scale + 2e-6, k
    sd = t.state_dict()
    mm = sd["UNet/down0/conv1/moving_mean"]
    assert mm.shape == (16,) and np.any(mm != 0)                    # moving statistics moved and are saved
    for _ in range(3):
        t.step(dev(x), dev(onehot), dev(wmap))
    assert np.isfinite(t.last_loss.item())


# ---- the same layer on bf16 activations (dtype 'bf16': configs 3-4's storage type) --------------------------------------
def _bf16(a):
    return torch.as_tensor(a, dtype=torch.float32).to(torch.bfloat16)


def _within_one_bf16_ulp(got, ref64, what, min_same=0.97):
    g = got.float().cpu().double()
    r = ref64.to(torch.bfloat16).double()
    bad = (g - ref64).abs() > torch.clamp(r.abs(), min=1e-30) * 2.0 ** -7 + 1e-6
    assert not bad.any(), "%s: %d values off by more than one bf16 ulp" % (what, int(bad.sum()))
    same = (g == r).double().mean().item()
    assert same > min_same, "%s: only %.4f bit-identical" % (what, same)


@pytest.mark.parametrize("shape", [(2, 16, 16, 16), (3, 9, 7, 64), (1, 5, 5, 256), (4, 33, 17, 8), (2, 64, 64, 32)])
def test_bn_bf16_stats_and_apply(shape):
    """sq_bn_stats_bf16 / sq_bn_apply_bf16 on bf16 operands against fp64 of the SAME (already rounded) values:
    statistics to f32 accuracy, act(x*scale+shift) equal to the fp64 value rounded once (1 ulp at rounding ties)."""
    from sequitr_amd import ops_bf16 as ob
    rng = np.random.default_rng(sum(shape))
    xb = _bf16(rng.standard_normal(shape) * 1.7 + 0.3)
    x64 = xb.double()
    C = shape[-1]
    mean, var = ob.bn_stats(xb.cuda())
    rmean = x64.reshape(-1, C).mean(0)
    rvar = ((x64.reshape(-1, C) - rmean) ** 2).mean(0)
    assert np.allclose(mean.cpu().numpy(), rmean.numpy(), rtol=2e-7, atol=1e-8)
    assert np.allclose(var.cpu().numpy(), rvar.numpy(), rtol=1e-6, atol=1e-9)
    scale = torch.tensor(rng.standard_normal(C).astype(np.float32))
    shift = torch.tensor(rng.standard_normal(C).astype(np.float32))
    for act in (None, "relu", "leaky"):
        y = ob.bn_apply(xb.cuda(), scale.cuda(), shift.cuda(), act)
        assert y.dtype == torch.bfloat16 and y.shape == xb.shape
        r = x64 * scale.double() + shift.double()
        r = torch.relu(r) if act == "relu" else (torch.where(r > 0, r, 0.2 * r) if act == "leaky" else r)
        _within_one_bf16_ulp(y, r, "apply %s" % act)


@pytest.mark.parametrize("act", [None, "relu"])
def test_bn_bf16_backward_vs_fp64_autograd(act):
    from sequitr_amd import ops_bf16 as ob
    rng = np.random.default_rng(1)
    xb = _bf16(rng.standard_normal((2, 12, 10, 32)) * 1.5 + 0.2)
    dyb = _bf16(rng.standard_normal(xb.shape))
    gamma, beta = (1 + 0.3 * rng.standard_normal(32)).astype(np.float32), rng.standard_normal(32).astype(np.float32)
    xt = xb.double().requires_grad_(True)
    gt = torch.tensor(gamma, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(beta, dtype=torch.float64, requires_grad=True)
    mu = xt.mean((0, 1, 2))
    var = ((xt - mu) ** 2).mean((0, 1, 2))
    pre = gt * (xt - mu) / torch.sqrt(var + 1e-3) + bt
    yt = torch.relu(pre) if act else pre
    (yt * dyb.double()).sum().backward()
    mean, v = ob.bn_stats(xb.cuda())
    scale, shift = ops.bn_fold(dev(gamma), dev(beta), mean, v, 1e-3)
    y = ob.bn_apply(xb.cuda(), scale, shift, act)
    # a pre-activation within f32 rounding of zero may land on the other side of the ReLU than in fp64: exclude it
    safe = (pre.detach().abs() > 1e-4) if act else torch.ones_like(pre, dtype=torch.bool)
    dx, dgamma, dbeta = ob.bn_bwd(xb.cuda(), dyb.cuda(), y, act, mean, v, dev(gamma), 1e-3)
    assert dx.dtype == torch.bfloat16
    assert bool(safe.all()) or safe.double().mean() > 0.999
    if bool(safe.all()):
        _within_one_bf16_ulp(dx, xt.grad, "dx", min_same=0.95)
        for got, ref, name in ((dgamma, gt.grad, "dgamma"), (dbeta, bt.grad, "dbeta")):
            ref = ref.numpy()
            assert np.max(np.abs(got.cpu().numpy() - ref)) <= 2e-5 * np.max(np.abs(ref)) + 1e-6, name


def _bn_state(w, rng):
    for k in [k for k in w if k.endswith("gamma")]:
        n = w[k].shape[0]
        w[k] = (1 + 0.2 * rng.standard_normal(n)).astype(np.float32)
        w[k[:-5] + "beta"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
        w[k[:-5] + "moving_mean"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
        w[k[:-5] + "moving_variance"] = (0.5 + rng.random(n)).astype(np.float32)
    return w


def test_unet_bf16_training_with_batchnorm_vs_emulation_and_fp64():
    """batch_norm=True through the bf16 graph: conv (no act, bf16 z) -> sq_bn_stats_bf16 -> fold -> sq_bn_apply_bf16
    (ReLU), backward through sq_bn_bwd_bf16.  Against oracle/bf16_ref.py with the same rounding points (forward) and
    the fp64 graph (gradients: as close to fp64 as the emulation is, the criterion of tests/test_gpu_bf16.py)."""
    from oracle import bf16_ref
    params = {"shape": (32, 32), "filters": (16, 32, 64), "batch_norm": True, "dropout": 0.0,
              "device": "cuda:0", "seed": 4, "dtype": "bf16"}
    rng = np.random.default_rng(6)
    x = tiles(6, 3, 32, 32)
    lab = rng.random((3, 32, 32)) < 0.4
    onehot = np.stack([~lab, lab], -1).astype(np.uint8)
    wmap = (1 + 3 * rng.random((3, 32, 32, 1))).astype(np.float32)
    t = UNetTrainer(params, learning_rate=0.01, warmup_steps=0)
    assert type(t.net).__name__ == "UNet2DBf16"
    assert any(k.endswith("gamma") for k in t.pbucket.names)
    t.load_state_dict(_bn_state(t.state_dict(), np.random.default_rng(2)))
    w0 = {k: v.copy() for k, v in t.state_dict().items()}
    loss = t.forward_backward(dev(x), dev(onehot), dev(wmap)).item()
    wt = {k: w0[k] for k in t.pbucket.names}
    rl, rg, rlogits = bf16_ref.unet_loss_and_grads_bf16(x, onehot, wmap, wt, params)
    l64, g64, _ = tr.unet_loss_and_grads(x, onehot, wmap, wt, params)
    logits = t.net.logits().detach().cpu().numpy()
    assert np.abs(logits - rlogits).max() <= 4 * 2.0 ** -7 * np.abs(rlogits).max()
    assert abs(loss - rl) <= 2e-3 * abs(rl) and abs(loss - l64) <= 0.02 * abs(l64)
    g = t.grads()
    for k in g64:
        tt, b, r = g64[k].ravel(), g[k].ravel().astype(np.float64), rg[k].ravel()
        nt = np.linalg.norm(tt)
        if nt < 1e-9:                   # a bias in front of a BN layer: exactly zero in fp64, the sum of the bf16
            assert np.linalg.norm(b) <= 3 * np.linalg.norm(r) + 1e-6, k     # roundings of dz here and in the emulation
            continue
        e_hip, e_emul = np.linalg.norm(b - tt) / nt, np.linalg.norm(r - tt) / nt
        cos = float(tt @ b / max(nt * np.linalg.norm(b), 1e-30))
        # BN's backward subtracts the batch projections of dy, which amplifies dy's bf16 rounding: the emulation is
        # 10-30 % away from fp64 in the deep layers of this random-init graph (0.3 % at the last block)
        assert e_hip <= 1.25 * e_emul + 0.005 and cos > 0.9, (k, e_hip, e_emul, cos)
    mm = t.state_dict()["UNet/down0/conv1/moving_mean"]
    assert np.any(mm != w0["UNet/down0/conv1/moving_mean"])          # moving statistics moved and are saved
    losses = [t.step(dev(x), dev(onehot), dev(wmap)).item() for _ in range(20)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


def test_unet_bf16_inference_with_batchnorm_vs_emulation():
    """Inference form (moving statistics folded into scale / shift): every level within the emulation's rounding, and the
    label image equal to the f32 graph's except where the two logits are within bf16 resolution of each other."""
    from oracle import bf16_ref
    from sequitr_amd.networks.unet import UNet2DBf16
    params = {"shape": (32, 32), "filters": (16, 32, 64), "batch_norm": True}
    w = _bn_state(init_unet_weights(params, 2), np.random.default_rng(3))
    net = UNet2DBf16(dict(params, device="cuda:0"), "infer")
    net.load_state_dict(w)
    x = tiles(5, 2, 32, 32)
    mask = net.predict(x).cpu().numpy()
    logits = net.logits().detach().float().cpu().numpy()
    rlogits = bf16_ref.unet_logits_bf16(x, w, dict(params, bn_moving=True))
    assert np.abs(logits - rlogits).max() <= 4 * 2.0 ** -7 * np.abs(rlogits).max()
    ref_logits = unet_oracle.unet_forward(x, w, params)
    ref_mask = unet_oracle.predict_mask(ref_logits)
    margin = np.abs(ref_logits[..., 1] - ref_logits[..., 0])
    differ = mask != ref_mask
    assert differ.mean() < 0.02 and np.all(margin[differ] < 0.05 * np.abs(ref_logits).max())
