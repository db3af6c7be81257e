"""GPU: the "mixed" convolutions (f32 tensors, bf16 multiply, f32 accumulate -- the GAN's dtype='bf16', BASELINE
config 5) vs an fp64 evaluation of the SAME bf16-rounded operands: products of bf16 numbers are exact in f32, so
only the f32 accumulation order differs (tolerance 2e-5 of the largest output); then the GAN losses and
gradients in mixed mode vs the fp64 graph of the unrounded operands (bf16 operand rounding: 3e-2 / cosine)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from oracle import torch_gan_ref as ref
from sequitr_amd import ops
from sequitr_amd.networks import gan
from tests.util import tiles, rand_weights

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bf16_round(a):
    return torch.as_tensor(a, dtype=torch.float32).to(torch.bfloat16).to(torch.float64)


def close(got, want, tol, what):
    want = np.asarray(want, dtype=np.float64)
    err = float(np.max(np.abs(np.asarray(got, dtype=np.float64) - want))) / max(float(np.max(np.abs(want))), 1e-30)
    assert err <= tol, "%s: rel err %.3g > %.3g" % (what, err, tol)


CASES = [(2, 32, 48, 16, 16, 3, "leaky", 1.0), (1, 32, 32, 32, 64, 3, "leaky", 0.37), (1, 16, 16, 128, 256, 3, None, 1.0),
         (2, 20, 27, 48, 20, 3, "relu", 1.0), (1, 40, 24, 8, 8, 3, "leaky", 0.5), (3, 17, 33, 8, 16, 3, None, 1.0),
         (1, 16, 16, 64, 32, 1, None, 1.0), (1, 16, 16, 8, 16, 1, "leaky", 2.0), (1, 8, 8, 512, 512, 3, "leaky", 0.05),
         (32, 4, 4, 64, 128, 3, "leaky", 1.0), (7, 8, 8, 16, 16, 3, None, 1.0)]


@pytest.mark.parametrize("N,H,W,Cin,Cout,K,act,wscale", CASES)
def test_mixed_conv_forward_and_dgrad(N, H, W, Cin, Cout, K, act, wscale):
    x, w, b = tiles(1, N, H, W, Cin), rand_weights(2, (K, K, Cin, Cout)), rand_weights(3, (Cout,), 0.1)
    ws = (torch.as_tensor(w) * np.float32(wscale)).numpy()           # the pack kernel rounds fl32(w * wscale) to bf16
    r = TF.conv2d(bf16_round(x).permute(0, 3, 1, 2), bf16_round(ws).permute(3, 2, 0, 1),
                  torch.as_tensor(b, dtype=torch.float64), padding=K // 2).permute(0, 2, 3, 1)
    r = TF.relu(r) if act == "relu" else (TF.leaky_relu(r, 0.2) if act == "leaky" else r)
    with ops.mixed_precision():
        got = ops.conv2d(dev(x), dev(w), dev(b), act=act, wscale=wscale)
    assert got.dtype == torch.float32
    close(got.cpu().numpy(), r.numpy(), 2e-5, "mixed conv")
    f32 = ops.conv2d(dev(x), dev(w), dev(b), act=act, wscale=wscale)    # outside the block: the exact-f32 kernel
    assert not torch.equal(f32, got)
    if Cout % 8 == 0:                                                   # dgrad = mixed conv with the transformed filter
        dy = tiles(4, N, H, W, Cout)
        xg = torch.zeros((N, Cin, H, W), dtype=torch.float64, requires_grad=True)
        TF.conv2d(xg, bf16_round(ws).permute(3, 2, 0, 1), padding=K // 2).backward(bf16_round(dy).permute(0, 3, 1, 2))
        with ops.mixed_precision():
            dx = ops.conv_dgrad_raw(dev(dy), dev(w), wscale)
        close(dx.cpu().numpy(), xg.grad.permute(0, 2, 3, 1).numpy(), 2e-5, "mixed dgrad")


@pytest.mark.parametrize("N,H,W,Cin,Cout,K", [(2, 32, 48, 16, 16, 3), (1, 32, 32, 32, 64, 3), (2, 21, 19, 16, 32, 3),
                                               (1, 16, 16, 64, 256, 1), (1, 16, 16, 128, 64, 3), (32, 4, 4, 64, 64, 3),
                                               (1, 24, 24, 32, 32, 1)])
def test_mixed_wgrad(N, H, W, Cin, Cout, K):
    x, dy = tiles(5, N, H, W, Cin), tiles(6, N, H, W, Cout)
    wt = torch.zeros((Cout, Cin, K, K), dtype=torch.float64, requires_grad=True)
    bt = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    TF.conv2d(bf16_round(x).permute(0, 3, 1, 2), wt, bt, padding=K // 2).backward(bf16_round(dy).permute(0, 3, 1, 2))
    with ops.mixed_precision():
        dw, db = ops.conv2d_wgrad(dev(x), dev(dy), K)
        dw2, _ = ops.conv2d_wgrad(dev(x), dev(dy), K)
    close(dw.cpu().numpy(), wt.grad.permute(2, 3, 1, 0).numpy(), 2e-5, "mixed dW")
    close(db.cpu().numpy(), bt.grad.numpy(), 2e-5, "mixed db")
    assert torch.equal(dw, dw2)                                         # fixed-order reduction


def test_layers_the_mixed_kernels_do_not_take_stay_f32():
    """image-side 1x1 convs (2 channels) and the first-layer shapes keep the exact-f32 kernels inside the block"""
    x, w = tiles(7, 2, 16, 16, 2), rand_weights(8, (1, 1, 2, 16))
    a = ops.conv2d(dev(x), dev(w), None, act="leaky")
    with ops.mixed_precision():
        b = ops.conv2d(dev(x), dev(w), None, act="leaky")
        assert ops.MIXED
    assert not ops.MIXED and torch.equal(a, b)


PARAMS = {"num_levels": 3, "batch_size": 4, "repeat_batch": 1, "num_epochs_per_level": 1, "learning_rate": 1e-3,
          "device": "cuda:0", "seed": 3, "num_batches_per_epoch": 2, "dtype": "bf16"}


@pytest.mark.parametrize("level,alpha", [(0, 1.0), (2, 0.4)])
def test_gan_losses_and_gradients_in_mixed_mode_vs_fp64(level, alpha):
    g = gan.GenerativeAdverserialNetwork(dict(PARAMS), mode=None)
    g.build()
    g.set_level(level)
    rng = np.random.default_rng(2)
    z = rng.standard_normal((4, 1, 1, 512)).astype(np.float32)
    x = rng.standard_normal((4,) + g.get_size(level) + (2,)).astype(np.float32)
    r = rng.random(4).astype(np.float32)
    d_vars, g_vars = g.get_training_variables(level)
    with g.precision():
        assert ops.MIXED
        _, d_loss, g_loss = g._build_network(dev(x), dev(z), alpha, r=dev(r))
        dg = torch.autograd.grad(d_loss, [v for _, v in d_vars], retain_graph=True, allow_unused=True)
        gg = torch.autograd.grad(g_loss, [v for _, v in g_vars], allow_unused=True)
    W = ref.to_torch(g.store.state_dict())
    _, rd, rg = ref.losses(torch.as_tensor(x, dtype=torch.float64), torch.as_tensor(z, dtype=torch.float64), alpha,
                           torch.as_tensor(r, dtype=torch.float64), W, g.filters, level)
    assert abs(d_loss.item() - rd.item()) <= 3e-2 * max(1.0, abs(rd.item()))
    assert abs(g_loss.item() - rg.item()) <= 3e-2 * max(1.0, abs(rg.item()))
    rdg = torch.autograd.grad(rd, [W[n] for n, _ in d_vars], retain_graph=True, allow_unused=True)
    rgg = torch.autograd.grad(rg, [W[n] for n, _ in g_vars], allow_unused=True)

    def cosine(ours, theirs):
        num = sum(float((a.double().cpu() * b).sum()) for a, b in zip(ours, theirs) if a is not None and b is not None)
        na = sum(float((a.double() ** 2).sum()) for a, b in zip(ours, theirs) if a is not None and b is not None)
        nb = sum(float((b ** 2).sum()) for a, b in zip(ours, theirs) if a is not None and b is not None)
        return num / np.sqrt(na * nb)

    assert cosine(dg, rdg) > 0.98 and cosine(gg, rgg) > 0.98


def test_gan_solver_steps_in_mixed_mode(tmp_path):
    g = gan.GenerativeAdverserialNetwork(dict(PARAMS, output=str(tmp_path / "o")), mode=None)
    g.build()
    g.set_level(2)
    rng = np.random.default_rng(4)
    z = dev(rng.standard_normal((4, 1, 1, 512)).astype(np.float32))
    x = dev(rng.standard_normal((4, 16, 16, 2)).astype(np.float32))
    before = g.store.state_dict()
    g.d_solver(x, z, 0.5)
    g.g_solver(x, z, 0.5)
    assert not ops.MIXED                                                # the flag does not leak out of a step
    after = g.store.state_dict()
    assert all(np.isfinite(v) for v in g.last_losses)
    assert any(np.abs(after[k] - before[k]).max() > 0 for k in before)
    img = g.predict(latent=np.zeros((2, 1, 1, 512), np.float32))
    assert tuple(img.shape) == (2, 16, 16, 2) and img.dtype == torch.float32


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(32, 4, 4, 512, 512), (32, 8, 8, 256, 256), (8, 4, 4, 64, 32), (5, 8, 8, 32, 64),
                                            (3, 4, 8, 16, 8)])
def test_mosaic_addressing_inside_the_mixed_conv_equals_pack_conv_unpack(N, H, W, Cin, Cout, monkeypatch):
    """sq_conv2d_nhwc_mixed_mosaic_f32 (the kernel addresses the compact small-image tensors through the mosaic map) against
    mosaic_pack -> conv -> mosaic_unpack: forward with bias + leaky, dgrad, and the act-gated dgrad, bit for bit."""
    x = torch.from_numpy(tiles(91, N, H, W, Cin)).cuda()
    w = torch.from_numpy(rand_weights(92, (3, 3, Cin, Cout), 0.05)).cuda()
    b = torch.from_numpy(rand_weights(93, (Cout,), 0.1)).cuda()
    dy = torch.from_numpy(tiles(94, N, H, W, Cout)).cuda()
    out = []
    for inside in (True, False):
        monkeypatch.setattr(ops, "MOSAIC_IN_KERNEL", inside)
        with ops.mixed_precision():
            y = ops.conv2d(x, w, b, act="leaky", wscale=0.7)
            dx = ops.conv_dgrad_raw(dy, w, 0.7)
            gated = ops.conv_dgrad_actgate(dy, w, 0.7, x, "leaky")
            if gated is None:                                   # the unfused pair the tape runs where the fused form does not exist
                gated = ops.act_bwd(dx, x, "leaky")
            dw, db = ops.conv2d_wgrad(x, dy, 3, want_bias=True, dw_scale=0.7) if (Cin % 16 == 0 and Cout % 16 == 0) else (y, y)
        out.append((y, dx, gated, dw, db))
    for a, bb in zip(out[0], out[1]):
        assert torch.equal(a, bb)
