"""GPU: the bf16 kernels vs an fp64 evaluation of the SAME bf16-rounded operands (the products of
bf16 numbers are exact in fp32, so only the accumulation order and the final rounding to bf16 can
differ: every output must be within one bf16 ulp of the reference and almost all bit-identical)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from sequitr_amd import ops_bf16 as ob
from tests.util import tiles, rand_weights

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def bf16_round(a):
    return torch.as_tensor(a, dtype=torch.float32).to(torch.bfloat16).to(torch.float64)


def check_bf16(got, ref64, what):
    """got: bf16 tensor; ref64: fp64 reference before the final rounding."""
    g = got.float().cpu().double()
    r = ref64.to(torch.bfloat16).double()
    ulp = torch.clamp(r.abs(), min=1e-30) * 2.0 ** -7
    bad = (g - ref64).abs() > ulp + 1e-6
    assert not bad.any(), "%s: %d values off by more than one bf16 ulp" % (what, int(bad.sum()))
    same = (g == r).double().mean().item()
    assert same > 0.97, "%s: only %.4f bit-identical" % (what, same)


CASES = [(2, 32, 48, 16, 16, 3, "relu"), (1, 32, 32, 16, 32, 3, "relu"), (1, 32, 32, 32, 32, 3, "relu"),
         (1, 16, 32, 64, 64, 3, "relu"), (1, 16, 16, 128, 256, 3, None), (2, 20, 27, 48, 16, 3, "leaky"),
         (1, 16, 16, 64, 32, 1, None), (1, 24, 24, 16, 64, 1, "relu"), (1, 8, 8, 256, 256, 3, "relu"),
         (2, 20, 27, 8, 8, 3, "leaky"), (1, 32, 32, 8, 16, 3, None), (1, 16, 16, 24, 32, 3, "relu"),   # KC = 8
         (1, 16, 16, 8, 32, 1, None)]


@pytest.mark.parametrize("N,H,W,Cin,Cout,K,act", CASES)
def test_conv_bf16(N, H, W, Cin, Cout, K, act):
    x, w = tiles(1, N, H, W, Cin), rand_weights(2, (K, K, Cin, Cout))
    b = rand_weights(3, (Cout,), 0.1)
    xb, wb = bf16_round(x), bf16_round(w)
    ref = TF.conv2d(xb.permute(0, 3, 1, 2), wb.permute(3, 2, 0, 1), torch.as_tensor(b, dtype=torch.float64), padding=K // 2)
    ref = ref.permute(0, 2, 3, 1)
    if act == "relu":
        ref = TF.relu(ref)
    elif act == "leaky":
        ref = TF.leaky_relu(ref, 0.2)
    wp = ob.pack_weights(dev(w))
    got = ob.conv2d(dev(x, torch.bfloat16), wp, dev(b), K, Cout, act=act)
    check_bf16(got, ref, "conv bf16 %s" % ((N, H, W, Cin, Cout, K, act),))


def test_dgrad_pack_equals_transposed_conv():
    N, H, W, Cin, Cout = 1, 16, 16, 32, 64                     # forward conv Cin -> Cout
    w, dy = rand_weights(4, (3, 3, Cin, Cout)), tiles(5, N, H, W, Cout)
    wb, dyb = bf16_round(w), bf16_round(dy)
    xg = torch.zeros((N, Cin, H, W), dtype=torch.float64, requires_grad=True)
    TF.conv2d(xg, wb.permute(3, 2, 0, 1), padding=1).backward(dyb.permute(0, 3, 1, 2))
    wp = ob.pack_weights(dev(w), transform=True)
    got = ob.conv2d(dev(dy, torch.bfloat16), wp, None, 3, Cin)
    check_bf16(got, xg.grad.permute(0, 2, 3, 1), "dgrad bf16")


@pytest.mark.parametrize("cin", [1, 2, 3, 7])
def test_first_conv_bf16(cin):
    x, w, b = tiles(6, 2, 40, 24, cin), rand_weights(7, (3, 3, cin, 16), 0.5), rand_weights(8, (16,), 0.1)
    ref = TF.relu(TF.conv2d(torch.as_tensor(x, dtype=torch.float64).permute(0, 3, 1, 2),
                            torch.as_tensor(w, dtype=torch.float64).permute(3, 2, 0, 1),
                            torch.as_tensor(b, dtype=torch.float64), padding=1)).permute(0, 2, 3, 1)
    check_bf16(ob.conv3x3_first(dev(x), dev(w), dev(b)), ref, "first conv")


@pytest.mark.parametrize("N,H,W,Cin,Cout,K", [(2, 32, 48, 16, 16, 3), (1, 32, 32, 32, 64, 3), (2, 21, 19, 16, 32, 3),
                                              (1, 16, 16, 64, 256, 1), (1, 16, 16, 128, 128, 3),
                                              # every channel-block shape of the dispatcher, ragged tiles, >1 tile per block
                                              (2, 20, 27, 32, 48, 3), (1, 17, 9, 16, 32, 3), (2, 12, 12, 64, 128, 1),
                                              (1, 7, 7, 32, 32, 1), (3, 40, 24, 48, 16, 3), (1, 9, 33, 16, 16, 1),
                                              (9, 32, 32, 32, 16, 3)])
def test_wgrad_bf16(N, H, W, Cin, Cout, K):
    x, dy = tiles(9, N, H, W, Cin), tiles(10, N, H, W, Cout)
    xb, dyb = bf16_round(x), bf16_round(dy)
    wt = torch.zeros((Cout, Cin, K, K), dtype=torch.float64, requires_grad=True)
    bt = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    TF.conv2d(xb.permute(0, 3, 1, 2), wt, bt, padding=K // 2).backward(dyb.permute(0, 3, 1, 2))
    dw, db = ob.conv2d_wgrad(dev(x, torch.bfloat16), dev(dy, torch.bfloat16), K)
    ref_w = wt.grad.permute(2, 3, 1, 0).numpy()                          # OIHW -> HWIO
    scale = np.abs(ref_w).max()
    assert np.abs(dw.cpu().numpy() - ref_w).max() <= 2e-6 * scale        # f32 accumulation of exact products
    assert np.abs(db.cpu().numpy() - bt.grad.numpy()).max() <= 2e-6 * np.abs(bt.grad.numpy()).max()
    dw2, _ = ob.conv2d_wgrad(dev(x, torch.bfloat16), dev(dy, torch.bfloat16), K)
    assert torch.equal(dw, dw2)


def test_streaming_ops_bf16():
    x = tiles(11, 2, 8, 12, 16)
    xb = bf16_round(x)
    got = ob.maxpool2x2(dev(x, torch.bfloat16)).float().cpu().double()
    assert torch.equal(got, TF.max_pool2d(xb.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1))
    dy = tiles(12, 2, 4, 6, 16)
    xt = xb.clone().permute(0, 3, 1, 2).requires_grad_(True)
    TF.max_pool2d(xt, 2, 2).backward(bf16_round(dy).permute(0, 3, 1, 2))
    got = ob.maxpool2x2_bwd(dev(x, torch.bfloat16), dev(dy, torch.bfloat16)).float().cpu().double()
    assert torch.equal(got, xt.grad.permute(0, 2, 3, 1))
    a, b, g = tiles(13, 1, 4, 4, 16), tiles(14, 1, 4, 4, 16), tiles(15, 1, 4, 4, 16)
    ab, bb, gb = bf16_round(a), bf16_round(b), bf16_round(g)
    for kind, f in (("eltwise_add", lambda p, q: p + q), ("eltwise_mul", lambda p, q: p * q), ("eltwise_sub", lambda p, q: p - q)):
        check_bf16(ob.bridge(dev(a, torch.bfloat16), dev(b, torch.bfloat16), kind), f(ab, bb), kind)
    da, db = ob.bridge_bwd(dev(g, torch.bfloat16), dev(a, torch.bfloat16), dev(b, torch.bfloat16), "eltwise_mul")
    check_bf16(da, gb * bb, "bridge bwd da")
    check_bf16(db, gb * ab, "bridge bwd db")
    y = tiles(16, 1, 4, 4, 16)
    got = ob.act_bwd(dev(g, torch.bfloat16), dev(y, torch.bfloat16), "relu").float().cpu().double()
    assert torch.equal(got, torch.where(bf16_round(y) > 0, gb, torch.zeros_like(gb)))
    xd = dev(np.ones((1, 32, 32, 16), np.float32), torch.bfloat16)
    y1, m1 = ob.dropout_fwd(xd, 0.4, seed=3)
    assert abs(m1.float().mean().item() - 0.6) < 0.02
    assert torch.equal(y1.float(), (m1.float() / 0.6).to(torch.bfloat16).float())
    assert torch.equal(ob.dropout_bwd(xd, m1, 0.4).float(), y1.float())
    f = dev(tiles(17, 1, 4, 4, 8))
    assert torch.equal(ob.to_f32(ob.to_bf16(f)), f.to(torch.bfloat16).float())
    s2d = ob.space_to_depth2(dev(tiles(18, 1, 4, 6, 8), torch.bfloat16)).float().cpu().numpy()
    ref = bf16_round(tiles(18, 1, 4, 6, 8)).numpy().reshape(1, 2, 2, 3, 2, 8).transpose(0, 1, 3, 2, 4, 5).reshape(1, 2, 3, 32)
    assert np.array_equal(s2d, ref)


@pytest.mark.parametrize("Cin,Cout", [(32, 16), (64, 32), (256, 128)])
@pytest.mark.parametrize("bridge", [None, "eltwise_mul", "eltwise_add"])
def test_convT_bf16(Cin, Cout, bridge):
    x, w, b = tiles(19, 2, 9, 13, Cin), rand_weights(20, (2, 2, Cout, Cin), 0.2), rand_weights(21, (Cout,), 0.1)
    skip = tiles(22, 2, 18, 26, Cout)
    up = TF.conv_transpose2d(bf16_round(x).permute(0, 3, 1, 2), bf16_round(w).permute(3, 2, 0, 1),
                             torch.as_tensor(b, dtype=torch.float64), stride=2).permute(0, 2, 3, 1)
    if bridge:
        u = up.to(torch.bfloat16).double()                         # the up-scaled value is stored as bf16 first
        s = bf16_round(skip)
        ref = u * s if bridge == "eltwise_mul" else u + s
    else:
        ref = up
    got = ob.convT2x2s2(dev(x, torch.bfloat16), ob.to_bf16(dev(w)), dev(b), skip=dev(skip, torch.bfloat16), bridge_kind=bridge)
    g = got.float().cpu().double()
    r = ref.to(torch.bfloat16).double()
    # a 1-ulp difference in the rounded up-scaled value can move the product by 1 ulp as well
    assert ((g - ref).abs() <= 2 * torch.clamp(ref.abs(), min=1e-30) * 2.0 ** -7 + 1e-6).all()
    assert (g == r).double().mean().item() > 0.97


def test_head_and_first_wgrad_bf16():
    x, w, b = tiles(23, 2, 24, 24, 16), rand_weights(24, (1, 1, 16, 2)), rand_weights(25, (2,), 0.1)
    xb = bf16_round(x)
    ref = (xb.reshape(-1, 16) @ torch.as_tensor(w, dtype=torch.float64).reshape(16, 2) + torch.as_tensor(b, dtype=torch.float64)).reshape(2, 24, 24, 2)
    logits, mask = ob.head_fwd(dev(x, torch.bfloat16), dev(w), dev(b))
    assert (logits.cpu().double() - ref).abs().max() < 1e-5
    assert torch.equal(mask.cpu(), ref.argmax(-1).to(torch.uint8)) or ((logits[..., 0] - logits[..., 1]).abs().cpu()[mask.cpu() != ref.argmax(-1).to(torch.uint8)] < 1e-5).all()
    dz = tiles(26, 2, 24, 24, 2)
    dx, dw, db = ob.head_bwd(dev(x, torch.bfloat16), dev(w), dev(dz))
    dzt = torch.as_tensor(dz, dtype=torch.float64).reshape(-1, 2)
    check_bf16(dx, (dzt @ torch.as_tensor(w, dtype=torch.float64).reshape(16, 2).T).reshape(2, 24, 24, 16), "head dx")
    assert np.allclose(dw.cpu().numpy().reshape(16, 2), (xb.reshape(-1, 16).T @ dzt).numpy(), rtol=1e-5, atol=1e-4)
    assert np.allclose(db.cpu().numpy(), dzt.sum(0).numpy(), rtol=1e-5, atol=1e-4)
    # first-layer weight gradient: f32 image (1..7 channels), bf16 dY
    for cin, cout in ((1, 16), (2, 16), (3, 32), (7, 16)):
        xi, dy = tiles(27, 2, 20, 28, cin), tiles(28, 2, 20, 28, cout)
        wt = torch.zeros((cout, cin, 3, 3), dtype=torch.float64, requires_grad=True)
        bt = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
        TF.conv2d(torch.as_tensor(xi, dtype=torch.float64).permute(0, 3, 1, 2), wt, bt, padding=1).backward(bf16_round(dy).permute(0, 3, 1, 2))
        dw, db = ob.conv3x3_first_wgrad(dev(xi), dev(dy, torch.bfloat16))
        assert np.allclose(dw.cpu().numpy(), wt.grad.permute(2, 3, 1, 0).numpy(), rtol=1e-5, atol=1e-4), cin
        assert np.allclose(db.cpu().numpy(), bt.grad.numpy(), rtol=1e-5, atol=1e-4), cin


def _batch(seed, n, size):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, size, size, 1)).astype(np.float32)
    lab = (rng.random((n, size, size)) < 0.3)
    return x, np.stack([~lab, lab], -1).astype(np.uint8), (1 + 9 * rng.random((n, size, size, 1))).astype(np.float32)


def test_unet_bf16_training_step_vs_cpu_emulation_and_f32():
    """The bf16 step against (a) oracle/bf16_ref.py, an fp64 emulation that rounds to bf16 at the same
    points (activations, filter copies, activation gradients), and (b) the fp64 graph without rounding.

    Forward: logits within 2 bf16 ulps of the emulation, loss within 1e-3 of it and 2 % of full precision.
    Backward: at random initialisation this 23-layer graph with multiplicative bridges is ill-conditioned
    for 8-bit mantissas -- activation gradients shrink by 1e4 and the emulation itself is 12-20 % away from
    the fp64 gradients, chaotically (two faithful bf16 evaluations differ by ~10 %).  The parity
    criterion is therefore: every weight gradient of the HIP path is as close to the fp64 truth as the
    reference emulation is (<= 1.25x its error + 0.5 %), with cosine > 0.98.  Each kernel alone is pinned
    to 1 ulp / 2e-6 by the tests above."""
    from oracle import bf16_ref
    from oracle import torch_ref as tr
    from sequitr_amd.train import UNetTrainer
    x, onehot, wmap = _batch(0, 2, 64)
    base = {"shape": (64, 64), "dropout": 0.0, "device": "cuda:0", "seed": 4}
    tb16 = UNetTrainer(dict(base, dtype="bf16"))
    assert type(tb16.net).__name__ == "UNet2DBf16"
    w0 = tb16.state_dict()
    l16 = tb16.forward_backward(dev(x), dev(onehot), dev(wmap)).item()
    rl, rg, rlogits = bf16_ref.unet_loss_and_grads_bf16(x, onehot, wmap, w0, base)
    l64, g64, _ = tr.unet_loss_and_grads(x, onehot, wmap, w0, base)
    logits = tb16.net.logits().detach().cpu().numpy()
    assert np.abs(logits - rlogits).max() <= 2 * 2.0 ** -7 * np.abs(rlogits).max()
    assert abs(l16 - rl) <= 1e-3 * abs(rl)
    assert abs(l16 - l64) <= 0.02 * abs(l64)
    g16 = tb16.grads()
    for k in g64:
        t, b, r = g64[k].ravel(), g16[k].ravel().astype(np.float64), rg[k].ravel()
        nt = max(np.linalg.norm(t), 1e-30)
        e_hip, e_emul = np.linalg.norm(b - t) / nt, np.linalg.norm(r - t) / nt
        cos = float(t @ b / max(nt * np.linalg.norm(b), 1e-30))
        assert e_hip <= 1.25 * e_emul + 0.005 and cos > 0.98, (k, e_hip, e_emul, cos)


def test_unet_bf16_trains_and_predicts():
    from sequitr_amd.train import UNetTrainer
    params = {"shape": (64, 64), "dropout": 0.2, "device": "cuda:0", "seed": 0, "filters": (16, 32, 64), "dtype": "bf16"}
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:64, 0:64]
    lab = ((yy - 32) ** 2 + (xx - 30) ** 2 < 200)
    x = (lab[None, ..., None] * 2.0 + rng.standard_normal((4, 64, 64, 1)) * 0.5).astype(np.float32)
    onehot = np.broadcast_to(np.stack([~lab, lab], -1)[None], (4, 64, 64, 2)).astype(np.uint8).copy()
    wmap = np.ones((4, 64, 64, 1), np.float32)
    t = UNetTrainer(params, learning_rate=0.003, warmup_steps=0)
    losses = [t.step(dev(x), dev(onehot), dev(wmap)).item() for _ in range(30)]
    assert losses[-1] < 0.5 * losses[0], losses
    from sequitr_amd.networks.unet import UNet2DBf16
    net = UNet2DBf16(dict(params), "infer")
    net.load_state_dict(t.state_dict())
    mask = net.predict(x).cpu().numpy()
    iou = np.logical_and(mask == 1, lab[None]).sum() / np.logical_or(mask == 1, lab[None]).sum()
    assert iou > 0.8


def test_fused_conv_block_backward_equals_the_unfused_tape():
    """UNet2DBf16.conv_block as one tape entry (dropout+ReLU backward in one pass, conv1's ReLU backward in the
    dgrad epilogue) gives the same loss and the same gradients, bit for bit, as the op-by-op tape."""
    from sequitr_amd.train import UNetTrainer
    base = {"shape": (64, 64), "dropout": 0.4, "device": "cuda:0", "seed": 5, "filters": (16, 32, 64), "dtype": "bf16"}
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 64, 64, 1)).astype(np.float32)
    lab = rng.random((2, 64, 64)) < 0.4
    onehot = np.stack([~lab, lab], -1).astype(np.uint8)
    wmap = (1 + rng.random((2, 64, 64, 1))).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to("cuda:0")
    a, b = UNetTrainer(dict(base, fuse_block=True)), UNetTrainer(dict(base, fuse_block=False))
    la = a.forward_backward(d(x), d(onehot), d(wmap))
    lb = b.forward_backward(d(x), d(onehot), d(wmap))
    assert la.item() == lb.item()
    ga, gb = a.grads(), b.grads()
    for k in gb:
        assert np.array_equal(ga[k], gb[k]), k


def test_overridden_wiring_switches_the_single_consumer_fusions_off():
    """the gate / handoff fusions promise that a block output has ONE differentiable consumer, which is the
    reference's wiring (unet.py:241-253), not a property of every subclass.  A net whose build() reads the last block's
    output twice (a second head: deep supervision) must get the full backward: loss and every gradient equal, bit for
    bit, to the op-by-op tape (fuse_block=False), and the fusions must report themselves off."""
    from sequitr_amd.train import UNetTrainer
    from sequitr_amd.networks.unet import UNet2DBf16

    class TwoHeads(UNet2DBf16):
        def build(self, features):
            logits = UNet2DBf16.build(self, features)
            with self.variable_scope('UNet'), self.variable_scope('to_image'):      # the same head variables again
                again = self.conv_layer_1x1(self._net[-2], self.n_outputs)
            return logits + 0.5 * again

    base = {"shape": (64, 64), "dropout": 0.4, "device": "cuda:0", "seed": 5, "filters": (16, 32, 64), "dtype": "bf16",
            "fuse_head_loss": False}
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 64, 64, 1)).astype(np.float32)
    lab = rng.random((2, 64, 64)) < 0.4
    onehot = np.stack([~lab, lab], -1).astype(np.uint8)
    wmap = (1 + rng.random((2, 64, 64, 1))).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to("cuda:0")
    a = UNetTrainer(dict(base, fuse_block=True), net_cls=TwoHeads)
    b = UNetTrainer(dict(base, fuse_block=False), net_cls=TwoHeads)
    assert not a.net._plain_wiring() and UNetTrainer(dict(base)).net._plain_wiring()
    la = a.forward_backward(d(x), d(onehot), d(wmap))
    lb = b.forward_backward(d(x), d(onehot), d(wmap))
    assert la.item() == lb.item()
    ga, gb = a.grads(), b.grads()
    for k in gb:
        assert np.array_equal(ga[k], gb[k]), k
    assert np.abs(ga["UNet/to_image/kernel"]).max() > 0


def test_pack_plan_equals_the_single_packs():
    """PackPlan (one launch per step) writes exactly what pack_weights / to_bf16 write one by one."""
    from sequitr_amd.train import UNetTrainer
    t = UNetTrainer({"shape": (32, 32), "device": "cuda:0", "seed": 1, "filters": (16, 32, 64), "dtype": "bf16"})
    assert t.pack_plan is not None and t.pack_plan.n > 20
    t.pack_plan.run()
    checked = 0
    for name in t.pbucket.names:
        leaf = t.net._vars[name]
        packs = getattr(leaf, "_sq_packs", None)
        if not packs:
            continue
        plain = leaf.detach().clone()                                  # no cache on the clone
        for key, got in packs.items():
            if key == "N":
                ref = ob.pack_weights(plain)
            elif key == "T":
                ref = ob.pack_weights(plain, transform=True)
            elif key == "cast":
                ref = ob.to_bf16(plain)
            else:
                Cout, Cin = plain.shape[2], plain.shape[3]
                ref = ob.pack_weights(plain.reshape(1, 1, 4 * Cout, Cin))
            assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), (name, key)
            checked += 1
    assert checked >= 20


@pytest.mark.parametrize("cfg", [{"filters": (16, 32), "num_outputs": 3}, {"filters": (16, 32), "num_outputs": 5},
                                 {"filters": (32, 64), "num_outputs": 2}, {"filters": (16, 32), "bridge": "eltwise_add"},
                                 {"filters": (16, 32), "num_inputs": 2}, {"filters": (32, 64), "num_inputs": 3}])
def test_bf16_training_other_configurations_match_the_emulation(cfg):
    """class counts up to 5 and other schedules / bridges through the bf16 graph (fused tape entries on):
    loss close to the fp64 graph's and gradients aligned with it."""
    from oracle import torch_ref as tr
    from sequitr_amd.train import UNetTrainer
    params = dict({"shape": (32, 32), "dropout": 0.0, "device": "cuda:0", "seed": 3, "dtype": "bf16"}, **cfg)
    nout = params.get("num_outputs", 2)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 32, 32, params.get("num_inputs", 1))).astype(np.float32)
    lab = rng.integers(0, nout, (2, 32, 32))
    onehot = (lab[..., None] == np.arange(nout)).astype(np.uint8)
    wmap = (1 + rng.random((2, 32, 32, 1))).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to("cuda:0")
    t = UNetTrainer(params, learning_rate=0.01)
    w0 = t.state_dict()
    loss = t.forward_backward(d(x), d(onehot), d(wmap))
    rloss, rgrads, _ = tr.unet_loss_and_grads(x, onehot, wmap, w0, params)
    assert abs(loss.item() - rloss) <= 2e-2 * abs(rloss)
    g = t.grads()
    num = sum(float((g[k].astype(np.float64) * rgrads[k]).sum()) for k in rgrads)
    den = np.sqrt(sum(float((g[k].astype(np.float64) ** 2).sum()) for k in rgrads) * sum(float((rgrads[k] ** 2).sum()) for k in rgrads))
    assert num / den > 0.97, num / den


@pytest.mark.parametrize("kind", ["eltwise_mul", "eltwise_add", "eltwise_sub"])
def test_convT_bridge_dual_output_equals_the_two_kernels(kind):
    """sq_convT2x2s2_bridge_both_fwd_bf16 (training form of the decoder junction): up and merged from one pass,
    bit-identical to convT followed by the bridge kernel."""
    x, skip = tiles(31, 2, 12, 20, 64), tiles(32, 2, 24, 40, 32)
    w, b = rand_weights(33, (2, 2, 32, 64), 0.2), rand_weights(34, (32,), 0.1)
    wb = dev(w).to(torch.bfloat16)
    up_ref = ob.convT2x2s2(dev(x, torch.bfloat16), wb, dev(b))
    merged_ref = ob.bridge(up_ref, dev(skip, torch.bfloat16), kind)
    up, merged = ob.convT2x2s2_bridge_both(dev(x, torch.bfloat16), wb, dev(b), dev(skip, torch.bfloat16), kind)
    assert torch.equal(up, up_ref) and torch.equal(merged, merged_ref)


@pytest.mark.parametrize("kind", ["eltwise_mul", "eltwise_add", "eltwise_sub"])
@pytest.mark.parametrize("shape", [(2, 24, 40, 32, 16), (1, 64, 64, 64, 32), (3, 16, 16, 128, 64), (1, 34, 30, 16, 16)])
def test_dgrad_with_junction_epilogue_equals_dgrad_then_bridge_backward(kind, shape):
    """sq_conv2d_nhwc_dgrad_junction_bf16: d_up (space-to-depth) and d_skip from the dgrad kernel's epilogue, bit for
    bit what the dgrad conv followed by sq_bridge_bwd_s2d_bf16 writes (ragged tiles included)."""
    N, H, W, Cd, Cm = shape                                     # dy has Cd channels, merged has Cm
    dy = dev(tiles(41, N, H, W, Cd), torch.bfloat16)
    up, skip = dev(tiles(42, N, H, W, Cm), torch.bfloat16), dev(tiles(43, N, H, W, Cm), torch.bfloat16)
    w = dev(rand_weights(44, (3, 3, Cm, Cd), 0.1))              # forward filter merged -> block
    wp_t = ob.pack_weights(w, transform=True)
    dm = ob.conv2d(dy, wp_t, None, 3, Cm)
    g_ref, ds_ref = ob.bridge_bwd_s2d(dm, up, skip, kind)
    keep = kind == "eltwise_mul"
    g, ds = ob.conv2d_dgrad_junction(dy, wp_t, up if keep else None, skip if keep else None, kind, 3, Cm)
    assert torch.equal(g, g_ref) and torch.equal(ds, ds_ref)


@pytest.mark.parametrize("bridge", ["eltwise_mul", "eltwise_add", "eltwise_sub"])
def test_junction_handoff_gives_the_same_gradients_as_the_standalone_pass(bridge, monkeypatch):
    """functional_bf16.JunctionHandoff on / off: same loss, same gradients, bit for bit."""
    from sequitr_amd.train import UNetTrainer
    from sequitr_amd import functional_bf16 as FB
    base = {"shape": (64, 64), "dropout": 0.4, "device": "cuda:0", "seed": 5, "filters": (16, 32, 64), "dtype": "bf16",
            "bridge": bridge}
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 64, 64, 1)).astype(np.float32)
    lab = rng.random((2, 64, 64)) < 0.4
    onehot = np.stack([~lab, lab], -1).astype(np.uint8)
    wmap = (1 + rng.random((2, 64, 64, 1))).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to("cuda:0")
    out = []
    for fuse in (True, False):
        monkeypatch.setattr(FB, "FUSE_JUNCTION", fuse)
        t = UNetTrainer(dict(base))
        loss = t.forward_backward(d(x), d(onehot), d(wmap))
        out.append((loss.item(), t.grads()))
    assert out[0][0] == out[1][0]
    for k in out[1][1]:
        assert np.array_equal(out[0][1][k], out[1][1][k]), k


@pytest.mark.parametrize("cfg", [{"filters": (16, 32, 64)}, {"filters": (32, 64), "num_outputs": 3},
                                 {"filters": (16, 32), "dropout": 0.0}])
def test_head_plus_loss_tape_entry_equals_head_then_loss(cfg):
    """sq_conv1x1_head_wce_{fwd,bwd}_bf16 (logits never stored) against head -> weighted softmax-CE -> head backward:
    same loss bits, same gradient bits, and the same after scaling the loss (a non-unit gradient arriving at it)."""
    from sequitr_amd.train import UNetTrainer
    base = dict({"shape": (64, 64), "dropout": 0.4, "device": "cuda:0", "seed": 5, "dtype": "bf16"}, **cfg)
    C = base.get("num_outputs", 2)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 64, 64, 1)).astype(np.float32)
    lab = rng.integers(0, C, (2, 64, 64))
    onehot = (lab[..., None] == np.arange(C)).astype(np.uint8)
    wmap = (1 + 4 * rng.random((2, 64, 64, 1))).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to("cuda:0")
    a, b = UNetTrainer(dict(base, fuse_head_loss=True)), UNetTrainer(dict(base, fuse_head_loss=False))
    la, lb = a.forward_backward(d(x), d(onehot), d(wmap)), b.forward_backward(d(x), d(onehot), d(wmap))
    assert la.item() == lb.item()
    ga, gb = a.grads(), b.grads()
    for k in gb:
        assert np.array_equal(ga[k], gb[k]), k
    # a scaled loss: the incoming gradient is a device scalar in both tapes
    from sequitr_amd import functional as F
    outs = []
    for t in (a, b):
        t.gbucket.flat.zero_()
        if t.pack_plan is not None:
            t.pack_plan.run()
        t.net.dropout_masks = None
        if t.fuse_head_loss:
            loss = t.net.build_loss(d(x), d(onehot), d(wmap))
        else:
            loss = F.weighted_softmax_cross_entropy(t.net.build(d(x)), d(onehot), d(wmap))
        (loss * 0.37).backward()
        outs.append(t.grads())
    for k in outs[1]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("rate", [0.0, 0.4])
@pytest.mark.parametrize("shape", [(2, 64, 64), (3, 40, 24), (1, 18, 50), (2, 512, 512)])
def test_first_block_in_one_launch_equals_the_two_kernels(shape, rate):
    """sq_conv3x3_first_block_dropout_pool_bf16 (down0's conv1 made per tile inside conv2's kernel) against
    sq_conv3x3_first_fwd_mask_bf16 -> sq_conv2d_nhwc_fwd_dropout_pool_bf16: y1, its sign mask, the block output and the pooled
    tensor, bit for bit -- ragged tiles and image borders included."""
    N, H, W = shape
    rng = np.random.default_rng(13)
    x = dev(rng.standard_normal((N, H, W, 1)).astype(np.float32))
    w1, b1 = dev(rand_weights(71, (3, 3, 1, 16), 0.5)), dev(rng.standard_normal(16).astype(np.float32) * 0.1)
    w2, b2 = dev(rand_weights(72, (3, 3, 16, 16), 0.1)), dev(rng.standard_normal(16).astype(np.float32) * 0.1)
    step = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    y1r, m1r = ob.conv3x3_first_mask(x, w1, b1)
    outr, poolr = ob.conv2d_dropout_pool(y1r, ob.pack_weights(w2), b2, 3, 16, 'relu', rate, seed=5, step_dev=step)
    assert ob.conv_first_block_takes(x, w1, w2)
    y1, m1, out, pool = ob.conv_first_block_dropout_pool(x, w1, b1, ob.pack_weights(w2), b2, rate, seed=5, step_dev=step)
    assert torch.equal(y1, y1r) and torch.equal(m1, m1r)
    assert torch.equal(out, outr) and torch.equal(pool, poolr)


def test_first_block_fusion_gives_the_same_training_step(monkeypatch):
    """functional_bf16.FUSE_FIRST on / off: same loss, same gradients, bit for bit."""
    from sequitr_amd.train import UNetTrainer
    from sequitr_amd import functional_bf16 as FB
    base = {"shape": (64, 64), "dropout": 0.4, "device": "cuda:0", "seed": 5, "filters": (16, 32, 64), "dtype": "bf16"}
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 64, 64, 1)).astype(np.float32)
    lab = rng.random((2, 64, 64)) < 0.4
    onehot = np.stack([~lab, lab], -1).astype(np.uint8)
    wmap = (1 + rng.random((2, 64, 64, 1))).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to("cuda:0")
    out, calls = [], []
    real = ob.conv_first_block_dropout_pool
    monkeypatch.setattr(ob, "conv_first_block_dropout_pool", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    for fuse in (True, False):
        monkeypatch.setattr(FB, "FUSE_FIRST", fuse)
        t = UNetTrainer(dict(base))
        loss = t.forward_backward(d(x), d(onehot), d(wmap))
        out.append((loss.item(), t.grads()))
    assert calls == [1]                                         # the fused form ran, once, in the first trainer only
    assert out[0][0] == out[1][0]
    for k in out[1][1]:
        assert np.array_equal(out[0][1][k], out[1][1][k]), k


@pytest.mark.parametrize("shape,C", [((2, 64, 64, 16), 2), ((3, 40, 24, 32), 3), ((4, 512, 512, 16), 2)])
def test_deferred_loss_is_the_forward_kernels_loss(shape, C):
    """functional_bf16.deferred_loss(): the loss the backward kernel leaves (sq_conv1x1_head_wce_bwd_loss_bf16) equals the
    forward kernel's bit for bit -- also past 2048 blocks, where threads walk several pixels -- and so do the gradients,
    with a non-unit gradient arriving at the loss; without gradients the forward kernel still runs."""
    from sequitr_amd import functional_bf16 as FB
    N, H, W, Cin = shape
    rng = np.random.default_rng(9)
    x = dev(tiles(61, N, H, W, Cin), torch.bfloat16)
    w = dev(rand_weights(62, (1, 1, Cin, C), 0.3))
    bias = dev(rng.standard_normal(C).astype(np.float32))
    lab = rng.integers(0, C, (N, H, W))
    onehot = dev((lab[..., None] == np.arange(C)).astype(np.uint8))
    wmap = dev((1 + 4 * rng.random((N, H, W, 1))).astype(np.float32))
    res = []
    calls = []
    real = ob.head_wce_fwd
    for defer in (False, True):
        xs, ws, bs = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        ob.head_wce_fwd = lambda *a, **k: (calls.append(defer), real(*a, **k))[1]
        try:
            with FB.deferred_loss(defer):
                loss = FB.conv1x1_head_loss(xs, ws, bs, onehot, wmap)
        finally:
            ob.head_wce_fwd = real
        (loss * 0.61).backward()
        res.append((loss.detach().clone(), xs.grad, ws.grad, bs.grad))
    assert calls == [False]                                     # the deferred form did not launch the forward kernel
    for u, v in zip(res[0], res[1]):
        assert torch.equal(u, v)
    with torch.no_grad(), FB.deferred_loss():
        assert torch.equal(FB.conv1x1_head_loss(x, w, bias, onehot, wmap), res[0][0])


@pytest.mark.parametrize("shape", [(2, 12, 20, 64, 32), (1, 16, 16, 32, 16), (1, 8, 8, 256, 128)])
def test_convT_wgrad_writes_the_transpose_conv_layouts(shape):
    """sq_convT2x2s2_wgrad_bf16 = the 1x1 wgrad of the space-to-depth form, with dW permuted to (2,2,Cout,Cin) and db
    summed over the four sub-pixel columns ((q0 + q1) + q2) + q3 by the finish kernel."""
    N, H, W, Cin, Cout = shape
    x, g = dev(tiles(51, N, H, W, Cin), torch.bfloat16), dev(tiles(52, N, H, W, 4 * Cout), torch.bfloat16)
    dwp, dbp = ob.conv2d_wgrad(x, g, 1)
    dw_ref = dwp.reshape(Cin, 2, 2, Cout).permute(1, 2, 3, 0).contiguous()
    b4 = dbp.reshape(4, Cout)
    db_ref = ((b4[0] + b4[1]) + b4[2]) + b4[3]
    dw, db = ob.convT_wgrad(x, g, Cout)
    assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref)
    sink = torch.zeros(2 * 2 * Cout * Cin, device="cuda")
    dw2, db2 = ob.convT_wgrad(x, g, Cout, want_bias=False, dw_out=sink)
    assert db2 is None and torch.equal(sink.view(2, 2, Cout, Cin), dw_ref)


@pytest.mark.parametrize("shape", [(2, 32, 48, 16, 16), (1, 64, 64, 32, 32), (1, 32, 32, 64, 64), (2, 16, 16, 128, 128),
                                   (1, 36, 20, 16, 32)])
@pytest.mark.parametrize("rate", [0.0, 0.4])
def test_conv_with_pooled_copy_equals_conv_then_maxpool(shape, rate):
    """sq_conv2d_nhwc_fwd_dropout_pool_bf16: y as the conv (+ dropout) kernel writes it, ypool = maxpool2x2(y), bit for bit
    (ragged tiles included)."""
    N, H, W, Cin, Cout = shape
    x = dev(tiles(61, N, H, W, Cin), torch.bfloat16)
    w, b = dev(rand_weights(62, (3, 3, Cin, Cout), 0.1)), dev(rand_weights(63, (Cout,), 0.1))
    wp = ob.pack_weights(w)
    y_ref = ob.conv2d_dropout(x, wp, b, 3, Cout, "relu", rate, seed=7) if rate > 0 else ob.conv2d(x, wp, b, 3, Cout, act="relu")
    y, yp = ob.conv2d_dropout_pool(x, wp, b, 3, Cout, "relu", rate, seed=7)
    assert torch.equal(y, y_ref) and torch.equal(yp, ob.maxpool2x2(y_ref))


@pytest.mark.parametrize("dropout", [0.4, 0.0])
def test_pool_from_the_block_epilogue_gives_the_same_step(dropout, monkeypatch):
    """FB.FUSE_POOL on / off: same loss, same gradients, bit for bit."""
    from sequitr_amd.train import UNetTrainer
    from sequitr_amd import functional_bf16 as FB
    base = {"shape": (64, 64), "dropout": dropout, "device": "cuda:0", "seed": 5, "filters": (16, 32, 64), "dtype": "bf16"}
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 64, 64, 1)).astype(np.float32)
    lab = rng.random((2, 64, 64)) < 0.4
    onehot = np.stack([~lab, lab], -1).astype(np.uint8)
    wmap = (1 + rng.random((2, 64, 64, 1))).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to("cuda:0")
    out = []
    for fuse in (True, False):
        monkeypatch.setattr(FB, "FUSE_POOL", fuse)
        t = UNetTrainer(dict(base))
        loss = t.forward_backward(d(x), d(onehot), d(wmap))
        out.append((loss.item(), t.grads()))
    assert out[0][0] == out[1][0]
    for k in out[1][1]:
        assert np.array_equal(out[0][1][k], out[1][1][k]), k


@pytest.mark.parametrize("shape", [(2, 32, 48, 16, 16), (1, 64, 64, 32, 32), (1, 32, 32, 64, 64), (2, 16, 16, 128, 128),
                                   (1, 36, 20, 32, 48), (1, 24, 24, 64, 16)])
def test_sign_mask_of_the_forward_conv_and_the_dgrad_gated_by_it(shape):
    """conv2d_mask: y as the plain conv writes it, mask bit (pixel, c) == (y > 0); conv2d_dgrad_mask == conv2d_dgrad_relu
    gated by the tensor, with and without the dropout scale (ragged tiles and partial channel blocks included)."""
    N, H, W, Cin, Cout = shape
    x = dev(tiles(71, N, H, W, Cin), torch.bfloat16)
    w, b = dev(rand_weights(72, (3, 3, Cin, Cout), 0.1)), dev(rand_weights(73, (Cout,), 0.1))
    wp = ob.pack_weights(w)
    y_ref = ob.conv2d(x, wp, b, 3, Cout, act="relu")
    y, m = ob.conv2d_mask(x, wp, b, 3, Cout)
    assert torch.equal(y, y_ref)
    bits = np.unpackbits(m.cpu().numpy().reshape(N, H, W, Cout // 8), axis=-1, bitorder="little")
    assert np.array_equal(bits.astype(bool), (y_ref.float() > 0).cpu().numpy())
    dy = dev(tiles(74, N, H, W, 32), torch.bfloat16)
    w2 = dev(rand_weights(75, (3, 3, Cout, 32), 0.1))          # the next conv Cout -> 32; its dgrad returns Cout channels
    wp_t = ob.pack_weights(w2, transform=True)
    for scale in (1.0, 1.0 / 0.6):
        assert torch.equal(ob.conv2d_dgrad_mask(dy, wp_t, m, 3, Cout, scale=scale),
                           ob.conv2d_dgrad_relu(dy, wp_t, y_ref, 3, scale=scale))


def test_first_conv_sign_mask():
    x = dev(tiles(81, 2, 32, 48, 1))
    w, b = dev(rand_weights(82, (3, 3, 1, 16), 0.3)), dev(rand_weights(83, (16,), 0.1))
    y_ref = ob.conv3x3_first(x, w, b, act="relu")
    y, m = ob.conv3x3_first_mask(x, w, b)
    assert torch.equal(y, y_ref)
    bits = np.unpackbits(m.cpu().numpy(), axis=-1, bitorder="little")
    assert np.array_equal(bits.astype(bool), (y_ref.float() > 0).cpu().numpy())


def test_mask_gate_gives_the_same_step(monkeypatch):
    """FB.FUSE_MASK on / off: same loss, same gradients, bit for bit."""
    from sequitr_amd.train import UNetTrainer
    from sequitr_amd import functional_bf16 as FB
    base = {"shape": (64, 64), "dropout": 0.4, "device": "cuda:0", "seed": 5, "filters": (16, 32, 64), "dtype": "bf16"}
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 64, 64, 1)).astype(np.float32)
    lab = rng.random((2, 64, 64)) < 0.4
    onehot = np.stack([~lab, lab], -1).astype(np.uint8)
    wmap = (1 + rng.random((2, 64, 64, 1))).astype(np.float32)
    d = lambda a: torch.from_numpy(a).to("cuda:0")
    out = []
    for fuse in (True, False):
        monkeypatch.setattr(FB, "FUSE_MASK", fuse)
        t = UNetTrainer(dict(base))
        loss = t.forward_backward(d(x), d(onehot), d(wmap))
        out.append((loss.item(), t.grads()))
    assert out[0][0] == out[1][0]
    for k in out[1][1]:
        assert np.array_equal(out[0][1][k], out[1][1][k]), k


def test_grouped_weight_gradients_equal_the_single_launches():
    """ob.deferred_wgrads(): the weight gradients of several layers from ONE grouped launch (+ one finish launch) per kernel
    shape (sq_conv2d_nhwc_wgrad_group_bf16) against the one-layer launches: the same products, each layer cut into fewer,
    longer blocks -- equal to f32 rounding of sums of ~1e5 terms, and run-to-run identical.  3x3 layers of three block shapes,
    a 1x1 layer and a transpose-conv layer in one queue."""
    from sequitr_amd import ops_bf16 as ob
    rng = np.random.default_rng(21)
    shapes = [(2, 32, 32, 32, 32, 3), (2, 16, 16, 64, 64, 3), (1, 32, 48, 16, 32, 3), (2, 16, 16, 128, 128, 3),
              (2, 32, 32, 16, 16, 3), (1, 16, 16, 64, 32, 1), (2, 8, 8, 256, 256, 3)]
    layers = []
    for (N, H, W, Cin, Cout, K) in shapes:
        x = dev(rng.standard_normal((N, H, W, Cin)).astype(np.float32), torch.bfloat16)
        dy = dev(rng.standard_normal((N, H, W, Cout)).astype(np.float32), torch.bfloat16)
        layers.append((x, dy, K))
    xt = dev(rng.standard_normal((2, 16, 16, 64)).astype(np.float32), torch.bfloat16)
    gt = dev(rng.standard_normal((2, 16, 16, 4 * 32)).astype(np.float32), torch.bfloat16)
    single = [ob.conv2d_wgrad(x, dy, K, want_bias=True) for x, dy, K in layers]
    single_t = ob.convT_wgrad(xt, gt, 32, want_bias=True)

    def grouped():
        outs = [(torch.zeros((K, K, x.shape[3], dy.shape[3]), device="cuda"), torch.zeros((dy.shape[3],), device="cuda"))
                for x, dy, K in layers]
        out_t = (torch.zeros((2, 2, 32, 64), device="cuda"), torch.zeros((32,), device="cuda"))
        with ob.deferred_wgrads() as q:
            for (x, dy, K), (dw, db) in zip(layers, outs):
                r = ob.conv2d_wgrad(x, dy, K, want_bias=True, dw_out=dw, db_out=db)
                assert r[0] is dw and r[1] is db
            ob.convT_wgrad(xt, gt, 32, want_bias=True, dw_out=out_t[0], db_out=out_t[1])
            assert len(q.items) == len(layers) + 1 and float(outs[0][0].abs().max()) == 0.0     # nothing has run yet
        return outs, out_t
    (outs, out_t), (outs2, out_t2) = grouped(), grouped()
    for (dw, db), (rw, rb), (dw2, db2) in zip(outs + [out_t], single + [single_t], outs2 + [out_t2]):
        assert torch.equal(dw, dw2) and torch.equal(db, db2)
        scale = float(rw.abs().max())
        assert float((dw - rw).abs().max()) <= 2e-6 * scale * np.sqrt(1e3) and float((db - rb).abs().max()) <= 1e-4 * float(rb.abs().max()) + 1e-4
    # a layer launched through a queue of ONE keeps its own plan: bit-identical to the single launch
    dw1, db1 = torch.zeros_like(single[0][0]), torch.zeros_like(single[0][1])
    with ob.deferred_wgrads():
        ob.conv2d_wgrad(layers[0][0], layers[0][1], 3, want_bias=True, dw_out=dw1, db_out=db1)
    assert torch.equal(dw1, single[0][0]) and torch.equal(db1, single[0][1])
