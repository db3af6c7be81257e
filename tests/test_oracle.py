"""CPU: pin oracle/sq_oracle.c (the fmaf-chain restatement) against an independent
fp64 torch-CPU implementation of the same ops (oracle/torch_ref.py).

The reference ships no tests or golden vectors for the network path,
so this cross-check -- two different code paths, different summation order, wider
type -- is what pins the restatement; tolerance 1e-5 abs on O(1) activations
(SURVEY.md 7 step 3).
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import torch_ref as tr
from oracle import unet_oracle
from sequitr_amd.networks.unet import init_unet_weights
from tests.util import tiles, rand_weights


@pytest.mark.parametrize("cin,cout,k,act", [(1, 16, 3, "relu"), (16, 16, 3, "relu"), (32, 64, 3, "relu"),
                                            (16, 2, 1, None), (8, 16, 3, "leaky"), (2, 8, 1, "leaky"),
                                            (64, 32, 3, None)])
def test_conv_matches_fp64(cin, cout, k, act):
    x = tiles(1, 2, 12, 20, cin)
    w = rand_weights(2, (k, k, cin, cout))
    b = rand_weights(3, (cout,), 0.1)
    y = co.conv2d(x, w, b, act=act, wscale=0.75)
    ref = tr.to_nhwc_np(tr.conv2d(tr.to_nchw(x), w, b, act=act, wscale=0.75))
    assert np.max(np.abs(y - ref)) < 1e-5


def test_conv_chain_order_is_chunk_tap_channel():
    """The contract order, restated in pure Python for one output element."""
    cin, cout = 32, 4
    x = tiles(4, 1, 5, 5, cin)
    w = rand_weights(5, (3, 3, cin, cout))
    y = co.conv2d(x, w, None, act=None)
    from fractions import Fraction
    yy, xx, o = 2, 3, 1
    acc = np.float32(0)
    for cc in range(0, cin, 16):
        for ky in range(3):
            for kx in range(3):
                for c in range(cc, cc + 16):
                    # fmaf = exact rational product-sum, rounded ONCE to float32
                    p = Fraction(float(w[ky, kx, c, o])) * Fraction(float(x[0, yy + ky - 1, xx + kx - 1, c])) \
                        + Fraction(float(acc))
                    acc = np.float32(p)
    assert y[0, yy, xx, o] == acc


def test_convT_matches_fp64():
    x = tiles(6, 2, 6, 10, 32)
    w = rand_weights(7, (2, 2, 16, 32), 0.2)
    b = rand_weights(8, (16,), 0.1)
    skip = tiles(9, 2, 12, 20, 16)
    for bridge in (None, "eltwise_add", "eltwise_mul", "eltwise_sub"):
        y = co.convT2x2s2(x, w, b, skip=skip, bridge=bridge)
        up = tr.convT2x2s2(tr.to_nchw(x), w, b)
        ref = tr.to_nhwc_np(tr.bridge_op(up, tr.to_nchw(skip), bridge))
        assert np.max(np.abs(y - ref)) < 1e-5, bridge


def test_pools_upsample_argmax():
    x = tiles(10, 2, 8, 12, 8)
    xt = tr.to_nchw(x, torch.float32)
    assert np.array_equal(co.maxpool2x2(x), tr.to_nhwc_np(torch.nn.functional.max_pool2d(xt, 2, 2)))
    assert np.allclose(co.avgpool2x2(x), tr.to_nhwc_np(torch.nn.functional.avg_pool2d(xt, 2, 2)), atol=1e-6)
    assert np.array_equal(co.upsample_nn2x(x), np.repeat(np.repeat(x, 2, 1), 2, 2))
    z = tiles(11, 1, 4, 4, 3)
    z[0, 0, 0] = [1.0, 1.0, 0.5]                 # tie -> lowest index
    m = co.argmax_u8(z)
    assert m[0, 0, 0] == 0 and np.array_equal(m, np.argmax(z, -1).astype(np.uint8))


def test_pixelnorm():
    x = tiles(12, 1, 4, 4, 32)
    ref = x / np.sqrt(np.mean(x.astype(np.float64) ** 2, -1, keepdims=True) + 1e-8)
    assert np.allclose(co.pixelnorm(x), ref, rtol=1e-6, atol=1e-6)


def test_wsoftmax_ce_matches_autograd():
    rng = np.random.default_rng(13)
    z = rng.standard_normal((2, 6, 6, 2)).astype(np.float32) * 3
    lab = rng.integers(0, 2, (2, 6, 6))
    y = np.stack([(lab == 0), (lab == 1)], -1).astype(np.uint8)
    w = (1 + 9 * rng.random((2, 6, 6, 1))).astype(np.float32)
    loss, dz = co.wsoftmax_ce(z, y, w)
    rloss, rdz = tr.wsoftmax_ce(z, y, w)
    assert abs(loss - rloss) < 1e-10
    assert np.max(np.abs(dz - rdz)) < 1e-7


def test_unet_forward_matches_fp64_64px():
    """Whole wiring (unet.py:224-322) at 64x64: C oracle vs fp64 torch, every layer."""
    params = {"shape": (64, 64)}
    w = init_unet_weights(params, seed=0)
    x = tiles(0, 2, 64, 64)
    logits, net = unet_oracle.unet_forward(x, w, params, return_net=True)
    rlogits, rnet = tr.unet_forward(x, w, params, return_net=True)
    assert logits.shape == (2, 64, 64, 2) and len(net) == len(rnet) == 10
    for i, (a, b) in enumerate(zip(net, rnet)):
        assert a.shape == b.shape
        assert np.max(np.abs(a - b)) < 1e-4 * max(1.0, float(np.max(np.abs(b)))), i
    assert np.max(np.abs(logits - rlogits)) < 1e-5 * max(1.0, float(np.abs(rlogits).max()))
    # masks agree except where the two logits are within rounding of each other
    m, rm = unet_oracle.predict_mask(logits), np.argmax(rlogits, -1).astype(np.uint8)
    diff = m != rm
    gap = np.abs(rlogits[..., 0] - rlogits[..., 1])
    assert (gap[diff] < 1e-5).all()


def test_batchnorm_oracle_matches_fp64():
    """BN restatement (SURVEY A.1 optional batch_norm): stats / fold / apply vs plain fp64 numpy."""
    rng = np.random.default_rng(4)
    x = (rng.standard_normal((3, 10, 6, 8)) * 2 + 0.7).astype(np.float32)
    gamma, beta = rng.standard_normal(8).astype(np.float32), rng.standard_normal(8).astype(np.float32)
    mean, var = co.bn_stats(x)
    x64 = x.reshape(-1, 8).astype(np.float64)
    assert np.allclose(mean, x64.mean(0), rtol=1e-6, atol=1e-7) and np.allclose(var, x64.var(0), rtol=1e-6)
    scale, shift = co.bn_fold(gamma, beta, mean, var, 1e-3)
    y = co.bn_apply(x, scale, shift, act="relu")
    ref = np.maximum(gamma * (x64 - x64.mean(0)) / np.sqrt(x64.var(0) + 1e-3) + beta, 0).reshape(x.shape)
    assert np.allclose(y, ref, rtol=1e-5, atol=1e-5)


def test_unet_oracle_with_batchnorm_identity_stats_is_scaled_plain_net():
    """moving_mean 0 / moving_variance 1 / gamma 1 / beta 0: BN is a division by sqrt(1+eps) per layer."""
    from sequitr_amd.networks.unet import init_unet_weights
    params = {"filters": (16, 32), "batch_norm": True}
    w = init_unet_weights(params, 1)
    assert "UNet/down0/conv1/gamma" in w and w["UNet/down0/conv1/gamma"].min() == 1.0
    x = np.random.default_rng(0).standard_normal((1, 16, 16, 1)).astype(np.float32)
    z = unet_oracle.unet_forward(x, w, params)
    plain = unet_oracle.unet_forward(x, {k: v for k, v in w.items() if k.split("/")[-1] in ("kernel", "bias")},
                                     {"filters": (16, 32)})
    assert np.allclose(z, plain / np.sqrt(1.001) ** 8, rtol=1e-4, atol=1e-6)     # mul bridge: s^4 (up) * s^2 (skip), then 2 more layers


def test_convT3x3s2_restatement_matches_torch_conv_transpose():
    """up_kernel=(3,3): zero insertion + rotated SAME conv == conv_transpose2d(k=3, s=2) cropped to 2H x 2W
    (TF SAME for the adjoint conv pads 0 before / 1 after)."""
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 5, 6, 8)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 4, 8)) * 0.2).astype(np.float32)           # TF layout (kh,kw,Cout,Cin)
    b = rng.standard_normal(4).astype(np.float32)
    y = unet_oracle.convT3x3s2(x, w, b)
    xt = torch.tensor(x, dtype=torch.float64).permute(0, 3, 1, 2)
    wt = torch.tensor(w, dtype=torch.float64).permute(3, 2, 0, 1)                # (Cin, Cout, kh, kw)
    ref = torch.nn.functional.conv_transpose2d(xt, wt, torch.tensor(b, dtype=torch.float64), stride=2)
    ref = ref[:, :, :10, :12].permute(0, 2, 3, 1).numpy()
    assert y.shape == (2, 10, 12, 4) and np.allclose(y, ref, rtol=1e-5, atol=1e-5)


def test_unet_oracle_concat_bridge_matches_fp64():
    """bridge = 'concat' (unet.py:196-197, tf.concat([upscale, skip], -1)): the C-oracle wiring (convT, numpy
    concatenate, conv over 2f input channels in the chunk / tap / channel chain order) vs the fp64 torch graph."""
    params = {"shape": (32, 32), "bridge": "concat", "filters": (16, 32, 64)}
    w = init_unet_weights(params, seed=1)
    assert w["UNet/up1/conv1/kernel"].shape == (3, 3, 64, 32)
    x = tiles(3, 2, 32, 32)
    logits, net = unet_oracle.unet_forward(x, w, params, return_net=True)
    rlogits, rnet = tr.unet_forward(x, w, params, return_net=True)
    assert len(net) == len(rnet) == 6
    for i, (a, b) in enumerate(zip(net, rnet)):
        assert a.shape == b.shape and np.max(np.abs(a - b)) < 1e-4 * max(1.0, float(np.max(np.abs(b)))), i
