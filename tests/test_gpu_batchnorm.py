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
