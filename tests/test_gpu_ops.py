"""GPU parity, op by op: HIP kernels (through the C-ABI) vs the C oracle on the same
seeded inputs.  Convolutions, pooling, transpose-conv+bridge, the 1x1 head and the
argmax mask are BIT-EXACT (the f32 MFMA is an fmaf chain in the contract order);
pixel-norm and the loss are tolerance checks (stated per test)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from sequitr_amd import ops
from tests.util import tiles, rand_weights, assert_bit_exact

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


CONV_CASES = [
    # N, H, W, Cin, Cout, K, act
    (2, 32, 48, 16, 16, 3, "relu"),      # level-0 shape class
    (1, 32, 32, 16, 32, 3, "relu"),
    (1, 32, 32, 32, 32, 3, "relu"),
    (1, 16, 32, 64, 64, 3, "relu"),
    (1, 16, 16, 128, 128, 3, "relu"),
    (1, 16, 16, 256, 256, 3, "relu"),    # 16 chunks
    (2, 20, 27, 16, 16, 3, "relu"),      # ragged: H, W not multiples of the 16x16 tile
    (1, 7, 5, 32, 64, 3, None),          # smaller than one tile
    (1, 32, 32, 1, 16, 3, "relu"),       # first layer, direct kernel
    (1, 19, 23, 1, 16, 3, "relu"),
    (1, 16, 16, 2, 8, 1, "leaky"),       # GAN from_image
    (1, 32, 32, 8, 8, 3, "leaky"),       # GAN 8-channel chunk
    (1, 32, 32, 8, 16, 3, "leaky"),
    (1, 16, 16, 16, 2, 1, None),         # to_image head via the generic entry
    (1, 8, 8, 512, 512, 3, "leaky"),     # GAN deep layer
    (1, 16, 16, 64, 32, 1, None),        # 1x1 through the MFMA path
    (3, 17, 33, 16, 24, 3, "relu"),      # Cout not a multiple of 16: masked channel tail in the epilogue
    (1, 40, 40, 48, 40, 3, "leaky"),     # three input chunks, ragged output block
    (5, 16, 16, 32, 20, 1, None),        # 1x1, odd batch, Cout tail
    (37, 16, 16, 16, 16, 3, "relu"),     # more tiles than one persistent block run divides evenly
    (1, 1, 100, 16, 16, 3, "relu"),      # a single row
    (1, 100, 1, 16, 32, 3, None),        # a single column
    (2, 21, 19, 2, 16, 3, "relu"),       # multi-channel tiles (num_inputs 2..7): direct kernel
    (1, 32, 32, 3, 32, 3, "relu"),
    (2, 16, 24, 4, 16, 3, "relu"),
    (1, 20, 28, 7, 24, 3, None),
]


@pytest.mark.parametrize("N,H,W,Cin,Cout,K,act", CONV_CASES)
def test_conv_bit_exact(N, H, W, Cin, Cout, K, act):
    x = tiles(100 + Cin, N, H, W, Cin)
    w = rand_weights(200 + Cout, (K, K, Cin, Cout))
    b = rand_weights(300, (Cout,), 0.1)
    wscale = 0.5 if act == "leaky" else 1.0
    ref = co.conv2d(x, w, b, act=act, wscale=wscale)
    got = ops.conv2d(dev(x), dev(w), dev(b), act=act, wscale=wscale).cpu().numpy()
    assert_bit_exact(got, ref, "conv %s" % ((N, H, W, Cin, Cout, K, act),))


def test_conv_no_bias_and_wscale_rounding():
    x = tiles(1, 1, 16, 16, 16)
    w = rand_weights(2, (3, 3, 16, 16))
    ws = float(np.sqrt(2.0 / (9 * 16)))        # gan.py:75-79 equalised-LR scale
    assert_bit_exact(ops.conv2d(dev(x), dev(w), None, act="leaky", wscale=ws).cpu().numpy(),
                     co.conv2d(x, w, None, act="leaky", wscale=ws), "wscale")


@pytest.mark.parametrize("C", [16, 32, 8])
def test_pools_bit_exact(C):
    x = tiles(3, 2, 24, 40, C)
    assert_bit_exact(ops.maxpool2x2(dev(x)).cpu().numpy(), co.maxpool2x2(x), "maxpool")
    assert_bit_exact(ops.avgpool2x2(dev(x)).cpu().numpy(), co.avgpool2x2(x), "avgpool")


@pytest.mark.parametrize("Cin,Cout", [(32, 16), (64, 32), (256, 128), (16, 8), (48, 24), (32, 20)])
@pytest.mark.parametrize("bridge", [None, "eltwise_add", "eltwise_mul", "eltwise_sub"])
def test_convT_bridge_bit_exact(Cin, Cout, bridge):
    x = tiles(4, 2, 9, 13, Cin)                 # P = 234 pixels: ragged vs the 64-pixel block
    w = rand_weights(5, (2, 2, Cout, Cin), 0.2)
    b = rand_weights(6, (Cout,), 0.1)
    skip = tiles(7, 2, 18, 26, Cout)
    ref = co.convT2x2s2(x, w, b, skip=skip, bridge=bridge)
    got = ops.convT2x2s2(dev(x), dev(w), dev(b), skip=dev(skip), bridge=bridge).cpu().numpy()
    assert_bit_exact(got, ref, "convT %s" % bridge)


@pytest.mark.parametrize("Cin,Cout,N,H,W", [(64, 32, 2, 192, 192), (128, 64, 1, 192, 192), (256, 128, 1, 72, 100)])
def test_convT_persistent_kernel_many_tiles_bit_exact(Cin, Cout, N, H, W):
    """sq_convt_f32_v2.hip (128 x 128 block tiles, channel-transposed operands, persistent blocks): more tiles than
    blocks (a block walks several tiles and prefetches across the seam), one / two / four row tiles per pixel tile, a
    ragged last pixel tile -- against the C oracle, bit for bit."""
    x = tiles(14, N, H, W, Cin)
    w = rand_weights(15, (2, 2, Cout, Cin), 0.2)
    b = rand_weights(16, (Cout,), 0.1)
    skip = tiles(17, N, 2 * H, 2 * W, Cout)
    for bridge in ("eltwise_mul", None):
        ref = co.convT2x2s2(x, w, b, skip=skip if bridge else None, bridge=bridge)
        got = ops.convT2x2s2(dev(x), dev(w), dev(b), skip=dev(skip) if bridge else None, bridge=bridge).cpu().numpy()
        assert_bit_exact(got, ref, "convT v2 %s" % bridge)


def test_bridge_standalone():
    a, b = tiles(8, 1, 8, 8, 16), tiles(9, 1, 8, 8, 16)
    for kind, f in (("eltwise_add", np.add), ("eltwise_mul", np.multiply), ("eltwise_sub", np.subtract)):
        assert_bit_exact(ops.bridge(dev(a), dev(b), kind).cpu().numpy(), f(a, b), kind)


def test_head_logits_and_mask_bit_exact():
    x = tiles(10, 2, 24, 24, 16)
    w = rand_weights(11, (1, 1, 16, 2))
    b = rand_weights(12, (2,), 0.1)
    x[0, 0, 0] = 0.0                              # logits tie (both = bias?) -> not a tie unless b equal
    b[1] = b[0]
    ref = co.conv2d(x, w, b, act=None)
    logits, mask = ops.conv1x1_argmax(dev(x), dev(w), dev(b))
    assert_bit_exact(logits.cpu().numpy(), ref, "head logits")
    assert ref[0, 0, 0, 0] == ref[0, 0, 0, 1]     # a genuine tie is present
    assert_bit_exact(mask.cpu().numpy(), co.argmax_u8(ref), "mask")
    assert mask[0, 0, 0].item() == 0              # ties -> lowest index
    assert_bit_exact(ops.argmax_u8(logits).cpu().numpy(), co.argmax_u8(ref), "argmax_u8")


def test_upsample_and_pixelnorm():
    x = tiles(13, 2, 6, 10, 32)
    assert_bit_exact(ops.upsample_nn2x(dev(x)).cpu().numpy(), co.upsample_nn2x(x), "upsample")
    for C in (8, 32, 512):
        x = tiles(14, 1, 5, 7, C)
        got = ops.pixelnorm(dev(x)).cpu().numpy()
        # lane-parallel sum order differs from the oracle's sequential one: 1e-6 relative
        assert np.allclose(got, co.pixelnorm(x), rtol=2e-6, atol=1e-7)


def test_wsoftmax_ce_loss_and_grad():
    rng = np.random.default_rng(15)
    z = (rng.standard_normal((3, 40, 40, 2)) * 4).astype(np.float32)
    lab = rng.integers(0, 3, (3, 40, 40))         # class 2 -> all-zero one-hot row (unet.py:396-398)
    y = np.stack([(lab == 0), (lab == 1)], -1).astype(np.uint8)
    w = (1 + 9 * rng.random((3, 40, 40, 1))).astype(np.float32)
    rloss, rdz = co.wsoftmax_ce(z, y, w)
    loss, dz = ops.wsoftmax_ce(dev(z), dev(y), dev(w))
    # f32 exp/log per pixel, fp64 accumulation: 1e-6 relative on the loss, 1e-6*max|w|/P on grads
    assert abs(loss.item() - rloss) <= 1e-6 * abs(rloss)
    assert np.max(np.abs(dz.cpu().numpy() - rdz)) <= 2e-6 * 10.0 / (3 * 40 * 40)
    loss2, none = ops.wsoftmax_ce(dev(z), dev(y), dev(w), want_grad=False)
    assert none is None and loss2.item() == loss.item()      # fixed-order reduction: reproducible


def test_errors_are_loud():
    from sequitr_amd._lib import SequitrHipError
    x = dev(tiles(1, 1, 16, 16, 12))
    w = dev(rand_weights(2, (3, 3, 12, 16)))
    with pytest.raises(SequitrHipError):
        ops.conv2d(x, w)                          # Cin = 12 unsupported
    with pytest.raises(ValueError):
        ops.conv2d(dev(tiles(1, 1, 16, 16, 16)), w)


@pytest.mark.parametrize("N,H,W,Cin,Cout,K", [(32, 4, 4, 64, 128, 3), (5, 8, 8, 32, 48, 3), (7, 4, 8, 16, 16, 3),
                                              (32, 1, 1, 528, 64, 1), (32, 4, 4, 64, 32, 1), (3, 5, 3, 16, 32, 3)])
def test_small_image_batches_run_as_mosaic_bit_exact(N, H, W, Cin, Cout, K):
    """GAN 4x4 / 8x8 levels and dense layers (gan.py:149-316): the batch is convolved as ONE mosaic image
    (3x3) or a flat pixel strip (1x1); every output keeps its exact fmaf chain."""
    x = tiles(N + H, N, H, W, Cin)
    w = rand_weights(Cin + Cout, (K, K, Cin, Cout))
    b = rand_weights(1, (Cout,), 0.1)
    if K == 3 and max(H, W) <= 8:
        assert ops._mosaic_plan(N, H, W) is not None
    y = ops.conv2d(dev(x), dev(w), dev(b), act="leaky", wscale=0.7)
    assert_bit_exact(y.cpu().numpy(), co.conv2d(x, w, b, act="leaky", wscale=0.7), "mosaic conv")
    # pack / unpack are inverse on the image cells and write zeros everywhere else
    if K == 3:
        R, Cc = ops._mosaic_plan(N, H, W)
        m = ops.mosaic_pack(dev(x), R, Cc)
        assert_bit_exact(ops.mosaic_unpack(m, N, H, W, R, Cc).cpu().numpy(), x, "unpack(pack)")
        assert float(m.abs().sum().item()) == pytest.approx(float(np.abs(x).astype(np.float64).sum()), rel=1e-5)
    # weight gradient over the mosaic == fp64 reference
    dy = tiles(9, N, H, W, Cout)
    dw, db = ops.conv2d_wgrad(dev(x), dev(dy), K)
    xt = torch.tensor(x, dtype=torch.float64).permute(0, 3, 1, 2)
    wt = torch.zeros((Cout, Cin, K, K), dtype=torch.float64, requires_grad=True)
    bt = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    (torch.nn.functional.conv2d(xt, wt, bt, padding=K // 2) * torch.tensor(dy, dtype=torch.float64).permute(0, 3, 1, 2)).sum().backward()
    ref = wt.grad.permute(2, 3, 1, 0).numpy()
    assert np.max(np.abs(dw.cpu().numpy() - ref)) <= 2e-5 * np.max(np.abs(ref))
    assert np.max(np.abs(db.cpu().numpy() - bt.grad.numpy())) <= 2e-5 * np.max(np.abs(bt.grad.numpy()))


@pytest.mark.parametrize("M,K,N,act", [(32, 8208, 512, "leaky"), (96, 8208, 512, "leaky"), (5, 1028, 70, None),
                                       (128, 2048, 256, "relu")])
def test_dense_split_reduction_vs_fp64(M, K, N, act):
    """tf.layers.dense of the discriminator (gan.py:226-237): split-reduction kernel vs fp64; also reached
    through conv2d for a (N,1,1,F) tensor and a 1x1 filter."""
    rng = np.random.default_rng(M + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64) + b
    if act == "leaky":
        ref = np.where(ref > 0, ref, 0.2 * ref)
    elif act == "relu":
        ref = np.maximum(ref, 0)
    y = ops.dense(dev(x), dev(w), dev(b), act=act).cpu().numpy()
    assert np.max(np.abs(y - ref)) <= 2e-5 * np.max(np.abs(ref))
    y2 = ops.conv2d(dev(x.reshape(M, 1, 1, K)), dev(w.reshape(1, 1, K, N)), dev(b), act=act).cpu().numpy()
    assert_bit_exact(y2.reshape(M, N), y, "conv2d routes dense layers to the same kernel")


def test_tensors_over_2gib_take_the_64bit_kernel_and_keep_the_bits():
    """>= 2 GiB operands cannot use 32-bit buffer offsets: the dispatcher falls back to the one-tile-per-block
    kernel with 64-bit addressing.  Batch independence gives the check: every image of the big batch equals the
    same image convolved on its own (which runs the pipelined kernel)."""
    N, H, W, C = 130, 512, 512, 16                              # 130 * 512 * 512 * 16 * 4 B = 2.03 GiB
    g = torch.Generator(device="cuda:0")
    g.manual_seed(0)
    x = torch.randn((N, H, W, C), device="cuda:0", generator=g)
    w = dev(rand_weights(5, (3, 3, C, C)))
    b = dev(rand_weights(6, (C,), 0.1))
    assert x.numel() * 4 >= 2 ** 31
    y = ops.conv2d(x, w, b, act="relu")
    for n in (0, 77, 129):
        one = ops.conv2d(x[n:n + 1].contiguous(), w, b, act="relu")
        assert torch.equal(y[n:n + 1], one), n
    del x, y
    torch.cuda.empty_cache()
