"""GPU: the progressive GAN's bf16-STORAGE forms (dtype 'bf16', BASELINE config 5): every feature map and feature-map
gradient is a bf16 tensor in HBM, parameters / images / losses are f32 (sequitr_amd/ops_gan_bf16.py, csrc/sq_gan_bf16.hip).

Kernel level: each op against fp64 arithmetic on the SAME (already bf16-rounded) operands -- the result must be the fp64
value rounded once (1 bf16 ulp allowed at rounding ties); fused forms (activation gates) against the two stored passes
they replace, bit for bit.  Network level: losses and parameter gradients of one WGAN-GP step against oracle/torch_gan_ref.py
(fp64) beside the f32-storage 'mixed' form, whose convolutions round the same operands to bf16."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_gan_ref as ref
from sequitr_amd import functional as F
from sequitr_amd import ops
from sequitr_amd import ops_gan_bf16 as gb
from sequitr_amd.networks import gan

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rb(rng, shape, scale=1.0, shift=0.0):
    """bf16-representable random tensor on the GPU and its fp64 copy"""
    t = torch.as_tensor(rng.standard_normal(shape) * scale + shift, dtype=torch.float32).to(BF)
    return t.cuda(), t.double()


def one_ulp(got, ref64, what, min_same=0.97):
    assert got.dtype == BF, what
    g = got.float().cpu().double()
    r = ref64.to(BF).double()
    bad = (g - ref64).abs() > torch.clamp(r.abs(), min=1e-30) * 2.0 ** -7 + 1e-7
    assert not bad.any(), "%s: %d values off by more than one bf16 ulp" % (what, int(bad.sum()))
    same = (g == r).double().mean().item()
    assert same > min_same, "%s: only %.4f bit-identical" % (what, same)


def leaky_gate(t64, g64, slope):
    """the stored two-rounding gate: t = bf16 value; gate > 0 ? t : bf16(t * slope)"""
    return torch.where(g64 > 0, t64, (t64.float() * np.float32(slope)).to(BF).double())


@pytest.mark.parametrize("shape", [(2, 6, 5, 8), (3, 4, 4, 16), (2, 3, 7, 32), (1, 5, 5, 64), (2, 4, 4, 128), (1, 4, 4, 512)])
def test_pixelnorm_bf16_all_orders(shape):
    rng = np.random.default_rng(sum(shape))
    xg, x = rb(rng, shape)
    gg, g = rb(rng, shape)
    vg, v = rb(rng, shape)
    C, eps = shape[-1], 1e-8
    xt, gt = x.clone().requires_grad_(True), g.clone().requires_grad_(True)
    y = xt * torch.rsqrt((xt * xt).mean(-1, keepdim=True) + eps)
    one_ulp(ops.pixelnorm(xg, eps), y.detach(), "pixel_norm")
    (dx,) = torch.autograd.grad(y, xt, gt, create_graph=True)
    one_ulp(ops.pixelnorm_bwd(xg, gg, eps), dx.detach(), "pixel_norm backward", 0.95)
    ddg, ddx = torch.autograd.grad(dx, [gt, xt], v)
    dg2, dx2 = ops.pixelnorm_bwd2(xg, gg, vg, eps)
    one_ulp(dg2, ddg, "pixel_norm 2nd order d/dg", 0.95)
    # dx2 is a sum of four terms that cancel: error relative to the tensor's scale, not to each value
    err = (dx2.float().cpu().double() - ddx).abs().max().item()
    assert err <= 2.0 ** -7 * ddx.abs().max().item(), ("pixel_norm 2nd order d/dx", err)
    # gate form == backward, stored, then act_bwd, stored
    for act, slope in (("leaky", 0.2), ("relu", 0.0)):
        fused = ops.pixelnorm_bwd(xg, gg, eps, act=act)
        unfused = ops.act_bwd(ops.pixelnorm_bwd(xg, gg, eps), xg, act)
        assert torch.equal(fused, unfused), act


@pytest.mark.parametrize("shape", [(2, 8, 6, 8), (3, 4, 4, 24), (1, 16, 16, 64)])
def test_pool_and_broadcast_bf16(shape):
    rng = np.random.default_rng(sum(shape) + 1)
    xg, x = rb(rng, shape)
    N, H, W, C = shape
    p = x.reshape(N, H // 2, 2, W // 2, 2, C)
    want = ((p[:, :, 0, :, 0] + p[:, :, 0, :, 1]) + (p[:, :, 1, :, 0] + p[:, :, 1, :, 1])) * 0.25
    one_ulp(ops.avgpool2x2(xg), want, "average pool")
    one_ulp(ops.sumpool2x2(xg, 1.0), want * 4, "sum pool")
    sg, s = rb(rng, (N, H // 2, W // 2, C))
    up = s.repeat_interleave(2, 1).repeat_interleave(2, 2)
    assert torch.equal(ops.broadcast2x2(sg, 1.0).float().cpu().double(), up)            # a copy: exact
    one_ulp(ops.broadcast2x2(sg, 0.25), up * 0.25, "scaled broadcast")
    for act in ("leaky", "relu"):
        fused = ops.broadcast2x2_act_bwd(sg, xg, 0.25, act)
        assert torch.equal(fused, ops.act_bwd(ops.broadcast2x2(sg, 0.25), xg, act)), act
    # adjoint pair through autograd (what the discriminator's down-sampling and its backward use)
    xa = xg.clone().requires_grad_(True)
    ya = F.avgpool2x2(xa)
    (ga,) = torch.autograd.grad(ya, xa, sg)
    assert ga.dtype == BF and torch.equal(ga, ops.broadcast2x2(sg, 0.25))
    leak = ops.act_fwd(xg, "leaky")
    one_ulp(leak, torch.where(x > 0, x, x.float().mul(np.float32(0.2)).double()), "leaky forward")


@pytest.mark.parametrize("C,Ci", [(8, 2), (16, 2), (64, 1), (512, 2), (32, 4)])
def test_image_side_convolutions_bf16(C, Ci):
    """from_image (f32 image -> bf16 features), to_image (features -> image) and their weight gradients"""
    rng = np.random.default_rng(C + Ci)
    N, H, W = 2, 8, 8
    img = torch.as_tensor(rng.standard_normal((N, H, W, Ci)), dtype=torch.float32)
    w_in = torch.as_tensor(rng.standard_normal((1, 1, Ci, C)), dtype=torch.float32)
    b_in = torch.as_tensor(rng.standard_normal(C) * 0.1, dtype=torch.float32)
    ws = float(np.sqrt(np.float32(2.0 / C)))
    with ops.mixed_precision(True, store_bf16=True):
        y = ops.conv2d(img.cuda(), w_in.cuda(), b_in.cuda(), act="leaky", wscale=ws)
    pre = img.double().reshape(-1, Ci) @ (w_in.reshape(Ci, C) * np.float32(ws)).double() + b_in.double()
    want = torch.where(pre > 0, pre, 0.2 * pre).reshape(N, H, W, C)
    assert y.dtype == BF
    err = (y.float().cpu().double() - want).abs()
    assert (err <= want.abs() * 2.0 ** -7 + 1e-6).all()
    fg, f = rb(rng, (N, H, W, C))
    w_out = torch.as_tensor(rng.standard_normal((1, 1, C, Ci)), dtype=torch.float32)
    b_out = torch.as_tensor(rng.standard_normal(Ci) * 0.1, dtype=torch.float32)
    z = ops.conv2d(fg, w_out.cuda(), b_out.cuda(), act=None, wscale=ws)
    wantz = (f.reshape(-1, C) @ (w_out.reshape(C, Ci) * np.float32(ws)).double() + b_out.double()).reshape(N, H, W, Ci)
    assert z.dtype == torch.float32
    assert np.allclose(z.cpu().numpy(), wantz.numpy(), rtol=2e-5, atol=2e-5 * float(wantz.abs().max()))
    # dgrad forms: the same two kernels with the transposed matrix
    with ops.mixed_precision(True, store_bf16=True):
        dxf = ops.conv_dgrad_raw(img.cuda(), w_out.cuda(), ws)          # d(to_image input): image grad -> features
    wantd = (img.double().reshape(-1, Ci) @ (w_out.reshape(C, Ci) * np.float32(ws)).double().t()).reshape(N, H, W, C)
    assert dxf.dtype == BF and ((dxf.float().cpu().double() - wantd).abs() <= wantd.abs() * 2.0 ** -7 + 1e-6).all()
    dimg = ops.conv_dgrad_raw(fg, w_in.cuda(), ws)                     # d(from_image input): features grad -> image
    wanti = (f.reshape(-1, C) @ (w_in.reshape(Ci, C) * np.float32(ws)).double().t()).reshape(N, H, W, Ci)
    assert dimg.dtype == torch.float32
    assert np.allclose(dimg.cpu().numpy(), wanti.numpy(), rtol=2e-5, atol=2e-5 * float(wanti.abs().max()))
    # weight / bias gradients
    dw, db = ops.conv_wgrad_raw(img.cuda(), fg, 1, want_bias=True, dw_scale=ws)
    assert np.allclose(dw.cpu().numpy().reshape(Ci, C), (img.double().reshape(-1, Ci).t() @ f.reshape(-1, C)).numpy() * ws,
                       rtol=1e-5, atol=1e-4)
    assert np.allclose(db.cpu().numpy(), f.reshape(-1, C).sum(0).numpy(), rtol=1e-5, atol=1e-4)
    dw2, db2 = ops.conv_wgrad_raw(fg, img.cuda(), 1, want_bias=True, dw_scale=ws)
    assert np.allclose(dw2.cpu().numpy().reshape(C, Ci), (f.reshape(-1, C).t() @ img.double().reshape(-1, Ci)).numpy() * ws,
                       rtol=1e-5, atol=1e-4)
    assert np.allclose(db2.cpu().numpy(), img.double().reshape(-1, Ci).sum(0).numpy(), rtol=1e-5, atol=1e-4)
    assert tuple(dw.shape) == (1, 1, Ci, C) and tuple(dw2.shape) == (1, 1, C, Ci)


def _conv_ref(x64, w, ws, bias=None, act=None):
    """fp64 conv of bf16-valued x with the bf16-rounded scaled filter (what the MFMA kernels multiply)"""
    wq = (w * np.float32(ws)).to(BF).double()
    y = torch.nn.functional.conv2d(x64.permute(0, 3, 1, 2), wq.permute(3, 2, 0, 1), None if bias is None else bias.double(),
                                   padding=w.shape[0] // 2).permute(0, 2, 3, 1)
    return torch.where(y > 0, y, 0.2 * y) if act == "leaky" else y


@pytest.mark.parametrize("N,H,W,Cin,Cout,K", [(6, 4, 4, 32, 16, 3), (5, 8, 8, 16, 32, 3), (2, 16, 16, 8, 16, 3),
                                              (2, 16, 16, 16, 8, 3), (1, 32, 32, 8, 8, 3), (4, 4, 4, 64, 32, 1),
                                              (8, 4, 4, 256, 64, 3), (4, 8, 8, 512, 32, 3)])   # the last two: split-K
def test_weighted_conv_bf16_forward_dgrad_gate_and_wgrad(N, H, W, Cin, Cout, K, monkeypatch):
    """3x3 / 1x1 feature convolutions on bf16 tensors: small-image batches through the mosaic addressing, 8-channel
    sides through the ragged weight-gradient form"""
    rng = np.random.default_rng(N * 1000 + H + Cin + Cout)
    xg, x = rb(rng, (N, H, W, Cin))
    w = torch.as_tensor(rng.standard_normal((K, K, Cin, Cout)), dtype=torch.float32)
    b = torch.as_tensor(rng.standard_normal(Cout) * 0.1, dtype=torch.float32)
    ws = float(np.sqrt(np.float32(2.0 / (K * K * Cout))))
    wd, bd = w.cuda(), b.cuda()
    y = ops.conv2d(xg, wd, bd, act="leaky", wscale=ws)
    one_ulp(y, _conv_ref(x, w, ws, b, "leaky"), "forward", 0.95)
    if W < 16:                                                  # the same bits without the mosaic / strip views ...
        monkeypatch.setattr(ops, "USE_MOSAIC", False)
        plain = ops.conv2d(xg, wd, bd, act="leaky", wscale=ws)
        monkeypatch.setattr(ops, "USE_MOSAIC", True)
        if Cin < 128:
            assert torch.equal(y, plain)
        else:                                                   # ... unless the mosaic launch split its reduction (other f32 order)
            assert not torch.equal(y, plain) or Cin < 128
            one_ulp(plain, _conv_ref(x, w, ws, b, "leaky"), "forward, unsplit", 0.95)
            assert (y.float() - plain.float()).abs().max().item() <= 2.0 ** -7 * plain.float().abs().max().item()
    dyg, dy = rb(rng, (N, H, W, Cout))
    dx = ops.conv_dgrad_raw(dyg, wd, ws)
    wt = torch.flip(w, (0, 1)).permute(0, 1, 3, 2).contiguous()
    one_ulp(dx, _conv_ref(dy, wt, ws), "dgrad", 0.95)
    fused = ops.conv_dgrad_actgate(dyg, wd, ws, xg, "leaky")
    if fused is not None:
        assert torch.equal(fused, ops.act_bwd(dx, xg, "leaky"))
    else:
        assert K == 1 and W < 16                                # the flat 1x1 strip has no gated form
    dw, db = ops.conv_wgrad_raw(xg, dyg, K, want_bias=True, dw_scale=ws)
    xp = torch.nn.functional.pad(x, (0, 0, K // 2, K // 2, K // 2, K // 2))
    want = torch.stack([torch.stack([(xp[:, ky:ky + H, kx:kx + W].reshape(-1, Cin).t() @ dy.reshape(-1, Cout))
                                     for kx in range(K)]) for ky in range(K)]) * ws
    scale = float(want.abs().max())
    assert np.abs(dw.cpu().numpy() - want.numpy()).max() <= 2e-5 * scale + 1e-6
    assert np.allclose(db.cpu().numpy(), dy.reshape(-1, Cout).sum(0).numpy(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 32, 32, 128, 64), (3, 64, 48, 64, 32), (2, 128, 128, 32, 16), (1, 256, 256, 16, 8),
                                             (2, 16, 16, 8, 8), (1, 40, 24, 16, 24)])
def test_pixel_norm_in_the_conv_epilogue(N, H, W, Cin, Cout):
    """sq_conv2d_nhwc_fwd_pixelnorm_bf16 (SURVEY 8b "pixelnorm epilogue flag"; gan.py:86-97 is conv -> bias -> activation ->
    pixel_norm as ONE op): y is the plain conv's y bit for bit, ynorm is the stand-alone pixel norm of that y up to the f32
    rounding of the per-pixel factor (the squares are added in another order: at most one bf16 ulp, nearly all identical) and
    one ulp from fp64 on the stored y; want_y=False writes the same ynorm; through the tape (F.conv2d(pixelnorm_eps=) ->
    F.pixel_norm) the forward launches ONE kernel and the gradients are those of the two-op graph."""
    rng = np.random.default_rng(N + H + Cin + Cout)
    xg, _ = rb(rng, (N, H, W, Cin))
    w = torch.as_tensor(rng.standard_normal((3, 3, Cin, Cout)), dtype=torch.float32).cuda()
    b = torch.as_tensor(rng.standard_normal(Cout) * 0.1, dtype=torch.float32).cuda()
    ws = float(np.sqrt(np.float32(2.0 / (9 * Cout))))
    assert ops.conv2d_pixelnorm_takes(xg, w)
    y_ref = ops.conv2d(xg, w, b, act="leaky", wscale=ws)
    n_ref = ops.pixelnorm(y_ref, 1e-8)
    y, yn = ops.conv2d_pixelnorm(xg, w, b, act="leaky", wscale=ws, eps=1e-8)
    assert torch.equal(y, y_ref)
    y64 = y.double().cpu()
    want = y64 * torch.rsqrt((y64 * y64).mean(-1, keepdim=True) + 1e-8)
    one_ulp(yn, want, "fused pixel norm vs fp64 on the stored y", 0.98)
    same = (yn == n_ref).double().mean().item()
    assert same > 0.995 and (yn.float() - n_ref.float()).abs().max().item() <= 2.0 ** -7 * n_ref.float().abs().max().item(), same
    none, yn2 = ops.conv2d_pixelnorm(xg, w, b, act="leaky", wscale=ws, eps=1e-8, want_y=False)
    assert none is None and torch.equal(yn2, yn)
    # through the tape
    calls = []
    orig = ops.pixelnorm
    try:
        ops.pixelnorm = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        xr = xg.clone().requires_grad_(True)
        wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        with ops.mixed_precision(True, store_bf16=True):
            out = F.pixel_norm(F.conv2d(xr, wr, br, act="leaky", wscale=ws, pixelnorm_eps=1e-8), 1e-8)
            assert not calls and torch.equal(out, yn)
            g = torch.as_tensor(rng.standard_normal(tuple(out.shape)), dtype=torch.float32).to(BF).cuda()
            gx, gw, gb_ = torch.autograd.grad(out, [xr, wr, br], g)
            x2 = xg.clone().requires_grad_(True)
            w2, b2 = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            out2 = F.pixel_norm(F.conv2d(x2, w2, b2, act="leaky", wscale=ws), 1e-8)
            assert len(calls) == 1
            hx, hw, hb = torch.autograd.grad(out2, [x2, w2, b2], g)
    finally:
        ops.pixelnorm = orig
    assert torch.equal(gx, hx) and torch.equal(gw, hw) and torch.equal(gb_, hb)     # the backward reads y only


def test_minibatch_stdev_on_bf16_features_equals_the_f32_kernels_round_the_casts():
    """sq_mbstd_map_{fwd,bwd,bwd2}_bf16: the statistic and its two derivatives straight from bf16 features -- the bits the f32
    kernels gave between a cast of x to f32 and casts of the gradients back to bf16"""
    rng = np.random.default_rng(17)
    for N, groups in ((8, 2), (6, 1)):
        xg, _ = rb(rng, (N, 4, 4, 64))
        dy = torch.as_tensor(rng.standard_normal((N, 16)), dtype=torch.float32).cuda()
        vg, _ = rb(rng, (N, 4, 4, 64))
        xf, vf = xg.float(), vg.float()
        assert torch.equal(ops.mbstd_map(xg, groups, 16), ops.mbstd_map(xf, groups, 16))
        dx = ops.mbstd_map_bwd(xg, dy, groups)
        assert dx.dtype == BF and torch.equal(dx, ops.mbstd_map_bwd(xf, dy, groups).to(BF))
        ddy, dx2 = ops.mbstd_map_bwd2(xg, dy, vg, groups)
        rdy, rx2 = ops.mbstd_map_bwd2(xf, dy, vf, groups)
        assert dx2.dtype == BF and torch.equal(ddy, rdy) and torch.equal(dx2, rx2.to(BF))


def test_head_concat_and_dense_gradient_sinks():
    """F.head_concat (the discriminator's cast + concat + flatten in one pass) against the framework ops it replaces, forward,
    backward and the second-order pass through its adjoint; and the dense layers' weight gradients written into the
    parameters' sinks (first contribution overwrites, later ones accumulate) against the framework's sum of three."""
    rng = np.random.default_rng(31)
    N, C = 6, 64
    conv_g, conv = rb(rng, (N, 4, 4, C))
    mb = torch.as_tensor(rng.standard_normal((N, 16)), dtype=torch.float32).cuda()
    flat = F.head_concat(conv_g, mb)
    want = torch.cat([conv_g.float(), mb.reshape(N, 4, 4, 1)], -1).reshape(N, 16 * (C + 1))
    assert flat.dtype == torch.float32 and torch.equal(flat, want)
    cg, mg = conv_g.clone().requires_grad_(True), mb.clone().requires_grad_(True)
    g = torch.as_tensor(rng.standard_normal((N, 16 * (C + 1))), dtype=torch.float32).cuda().requires_grad_(True)
    dconv, dmb = torch.autograd.grad(F.head_concat(cg, mg), [cg, mg], g, create_graph=True)
    gv = g.detach().reshape(N, 4, 4, C + 1)
    assert dconv.dtype == BF and torch.equal(dconv, gv[..., :C].to(BF)) and torch.equal(dmb, gv[..., C].reshape(N, 16))
    v1 = torch.as_tensor(rng.standard_normal((N, 4, 4, C)), dtype=torch.float32).to(BF).cuda()
    v2 = torch.as_tensor(rng.standard_normal((N, 16)), dtype=torch.float32).cuda()
    (dg,) = torch.autograd.grad([dconv, dmb], [g], [v1, v2])    # d/d(dflat) of the split = the concat of the two cotangents
    assert torch.equal(dg, torch.cat([v1.float(), v2.reshape(N, 4, 4, 1)], -1).reshape(N, -1))
    # dense sinks: three contributions to one (Kin, N) kernel and its bias
    M, Kin, Nout = 8, 1040, 64
    w = torch.as_tensor(rng.standard_normal((Kin, Nout)) * 0.05, dtype=torch.float32).cuda().requires_grad_(True)
    b = torch.zeros(Nout, device="cuda").requires_grad_(True)
    xs = [torch.as_tensor(rng.standard_normal((M, Kin)), dtype=torch.float32).cuda() for _ in range(3)]

    def loss():
        return sum((F.dense(x, w, b, act="leaky") * (k + 1.0)).sum() for k, x in enumerate(xs))
    ref_w, ref_b = torch.autograd.grad(loss(), [w, b])
    sinks = {id(w): torch.full((Kin, Nout), 7.0, device="cuda"), id(b): torch.full((Nout,), 7.0, device="cuda")}
    with F.grad_sinks(sinks) as sk:
        got = torch.autograd.grad(loss(), [w, b], allow_unused=True)
    assert got[0] is None and got[1] is None and sk.touched == {id(w), id(b)}
    assert torch.allclose(sinks[id(w)], ref_w, rtol=1e-5, atol=1e-5) and torch.allclose(sinks[id(b)], ref_b, rtol=1e-5, atol=1e-4)


def _rel(a, t):
    a, t = np.asarray(a, np.float64), np.asarray(t, np.float64)
    return float(np.linalg.norm((a - t).ravel()) / max(np.linalg.norm(t.ravel()), 1e-30))


# Absolute bounds of the HIP path against the rounding-point emulation (oracle/gan_bf16_ref.py: float64 arithmetic, bf16
# roundings where the kernels store / round).  What is left between the two is f32 accumulation order and the rare stored
# value that lands on the other side of a bf16 rounding boundary because of it -- amplified by the same factor that turns
# the 2^-9 roundings themselves into 5-30 % of a gradient at random initialisation (printed beside it: fp64 vs emulation).
GRAD_BOUND = 2e-2        # norm-wise, every parameter gradient of d_loss and g_loss
LOSS_BOUND = 1e-3        # relative to max(1, |loss|)


def _whole_graph(level, alpha, levels, nb, penalty_active=True):
    """d_loss, g_loss and every parameter gradient of both, HIP (tape and solver path) vs the emulation vs fp64"""
    from tests.test_gpu_gan import make_gan, dev
    from oracle import gan_bf16_ref as emu
    rng = np.random.default_rng(2)
    g = make_gan(dtype="bf16", num_levels=levels, batch_size=nb)
    g.set_level(level)
    # the one-sided penalty (and with it the whole second-order pass) is exactly zero while |dD(mix)/dmix| < 1, which is
    # where a freshly initialised deep discriminator sits: scale its last layer so that the penalty is active on every
    # sample (asserted below on the emulation's gradient norms)
    z = rng.standard_normal((nb, 1, 1, 512)).astype(np.float32)
    r = rng.random(nb).astype(np.float32)
    x = rng.standard_normal((nb,) + g.get_size(level) + (2,)).astype(np.float32)
    t64 = lambda a: torch.as_tensor(a, dtype=torch.float64)
    det = {}
    emu.losses(t64(x), t64(z), alpha, t64(r), emu.to_torch(g.store.state_dict()), g.filters, level, details=det)
    gain = 3.0 / float(det["grad_norm"].min()) if penalty_active else 1.0   # |dD/dmix| is linear in the last layer's kernel
    with torch.no_grad():
        g.store.vars["GAN/discriminator/output/logits/kernel"].mul_(gain)
    ops.invalidate_packs()
    d_vars, g_vars = g.get_training_variables(level)
    names = [n for n, _ in d_vars + g_vars]
    with g.precision():
        _, d_loss, g_loss = g._build_network(dev(x), dev(z), alpha, r=dev(r))
        dg = torch.autograd.grad(d_loss, [v for _, v in d_vars], retain_graph=True, allow_unused=True)
        gg = torch.autograd.grad(g_loss, [v for _, v in g_vars], allow_unused=True)
    assert all(t.dtype == torch.float32 for t in dg + gg if t is not None)
    tape = (d_loss.item(), g_loss.item(), [None if t is None else t.cpu().numpy() for t in dg + gg])
    with g.precision(), F.fuse_act_gates(g._act_gates()):      # what d_solver / g_solver run in front of Adam
        g._pack_filters()
        _, sdg, (sd_loss, _) = g._d_grads(dev(x), dev(z), alpha, dev(r))
        sdg = [None if t is None else t.clone() for t in sdg]
        _, sgg, (sg_loss,) = g._g_grads(dev(x), dev(z), alpha)
    solver = (sd_loss.item(), sg_loss.item(), [None if t is None else t.cpu().numpy() for t in list(sdg) + list(sgg)])

    sd = g.store.state_dict()
    W = emu.to_torch(sd)
    det = {}
    _, ed, eg = emu.losses(t64(x), t64(z), alpha, t64(r), W, g.filters, level, details=det)
    print("level %d: logits kernel x %.1f, |dD(mix)/dmix| per sample %s" % (level, gain, np.round(det["grad_norm"].numpy(), 3)))
    if penalty_active:
        assert float(det["grad_norm"].min()) > 1.05, "the penalty is not active on every sample: the second-order pass is not exercised"
    edg = torch.autograd.grad(ed, [W[n] for n, _ in d_vars], retain_graph=True, allow_unused=True)
    egg = torch.autograd.grad(eg, [W[n] for n, _ in g_vars], allow_unused=True)
    want = [None if t is None else t.numpy() for t in edg + egg]
    W64 = ref.to_torch(sd)
    _, rd, rg = ref.losses(t64(x), t64(z), alpha, t64(r), W64, g.filters, level)
    rdg = torch.autograd.grad(rd, [W64[n] for n, _ in d_vars], retain_graph=True, allow_unused=True)
    rgg = torch.autograd.grad(rg, [W64[n] for n, _ in g_vars], allow_unused=True)
    truth = [None if t is None else t.numpy() for t in rdg + rgg]

    print("level %d alpha %.1f batch %d: d_loss fp64 %.6f emu %.6f hip %.6f (solver %.6f) | g_loss fp64 %.6f emu %.6f hip %.6f"
          % (level, alpha, nb, rd.item(), ed.item(), tape[0], solver[0], rg.item(), eg.item(), tape[1]))
    rounding, gap_tape, gap_solver = [], [], []
    for n, t, w, a, b_ in zip(names, truth, want, tape[2], solver[2]):
        assert (w is None) == (a is None) == (b_ is None), n
        if w is None:
            continue
        rounding.append(_rel(w, t)), gap_tape.append(_rel(a, w)), gap_solver.append(_rel(b_, w))
        print("  %-52s rounding alone (fp64 vs emu) %.4f | hip vs emu: tape %.5f solver %.5f"
              % (n, rounding[-1], gap_tape[-1], gap_solver[-1]))
    print("  mean / max: rounding alone %.4f / %.4f | hip vs emu tape %.5f / %.5f, solver %.5f / %.5f"
          % (np.mean(rounding), max(rounding), np.mean(gap_tape), max(gap_tape), np.mean(gap_solver), max(gap_solver)))
    return {"d": (rd.item(), ed.item(), tape[0], solver[0]), "g": (rg.item(), eg.item(), tape[1], solver[1]),
            "names": names, "rounding": rounding, "tape": gap_tape, "solver": gap_solver}


@pytest.mark.parametrize("level,alpha", [(0, 1.0), (1, 0.7), (2, 0.4), (2, 1.0)])
def test_losses_and_gradients_bf16_vs_the_rounding_point_emulation(level, alpha):
    """config 5 in ITS dtype against an oracle, whole graph -- d_loss, g_loss and every parameter
    gradient of both with the penalty ACTIVE (so the second-order pass carries weight), ABSOLUTE bounds.  Two evaluations
    of the HIP path: plain autograd over the tape, and the solver's own gradient path (_d_grads / _g_grads: activation
    gates, parameter-gradient sinks, grouped weight-gradient launches, the stacked D(Gz | X) pass) -- the one training runs.
    Measured on MI355X (round 4): HIP vs emulation mean 3e-4 .. 9e-4, max 4.5e-3 per gradient, losses to 5 digits, while
    the roundings alone move the same gradients by 0.17 - 0.28 (mean) and up to 0.64 -- the errors round 3 saw against
    fp64 ARE the bf16 roundings.  (Level 6 is tested below: there a whole-graph comparison cannot be tight, for a
    reason that is measured, not assumed.)"""
    res = _whole_graph(level, alpha, 3, 4)
    for k in ("d", "g"):
        _, e, t, s_ = res[k]
        assert abs(t - e) <= LOSS_BOUND * max(1.0, abs(e)) and abs(s_ - e) <= LOSS_BOUND * max(1.0, abs(e)), (k, res[k])
    worst = max(zip(res["tape"] + res["solver"], res["names"] + res["names"]))
    assert worst[0] <= GRAD_BOUND, worst
    assert np.mean(res["rounding"]) > 10 * np.mean(res["tape"])  # the gap to fp64 is the roundings, not the kernels


class _Replay(object):
    """oracle/gan_bf16_ref.OBSERVER: every leaf evaluation of the emulation -- forward pass, create_graph backward, second-order
    backward -- is run again through the HIP operator the product dispatches it to, ON THE EMULATION'S OWN OPERANDS, and
    compared with the value the emulation stored.  Operands are identical on both sides, so nothing compounds from layer to
    layer: a stored bf16 value may differ by one ulp where the f32 accumulation lands across a rounding boundary (a
    fraction of a per cent of the values), an f32 result by f32 accumulation error -- anything more is a defect in that
    kernel or a rounding point the emulation has in another place."""

    def __init__(self):
        self.count, self.worst = {}, {}

    @staticmethod
    def put(t, s):
        t = t.detach()
        return (t.to(torch.float32).to(BF) if s == 'b' else t.to(torch.float32)).contiguous().cuda()

    def same(self, kind, got, want, s, what):
        self.count[kind] = self.count.get(kind, 0) + 1
        g = got.detach().float().cpu().double().reshape(want.shape)
        if s == 'b':
            assert got.dtype == BF, (kind, what)
            # one ulp of the stored value, plus the f32 accumulation error of a sum that cancels to (nearly) nothing
            bad = (g - want).abs() > want.abs() * 2.0 ** -7 + 1e-5 * float(want.abs().max())
            frac_same = float((g == want).double().mean())
            assert not bad.any(), "%s %s: %d values off by more than one bf16 ulp" % (kind, what, int(bad.sum()))
            assert frac_same >= 0.97, "%s %s: only %.4f of the stored values identical" % (kind, what, frac_same)
            self.worst[kind + ' (bf16: identical fraction)'] = min(self.worst.get(kind + ' (bf16: identical fraction)', 1.0), frac_same)
        else:
            assert got.dtype == torch.float32, (kind, what)
            err = float((g - want).norm() / want.norm().clamp_min(1e-30))
            assert err <= 5e-5, "%s %s: f32 result off by %.3e (norm-wise)" % (kind, what, err)
            self.worst[kind + ' (f32: norm-wise error)'] = max(self.worst.get(kind + ' (f32: norm-wise error)', 0.0), err)

    def __call__(self, kind, a, out):
        put = self.put
        if kind in ('conv', 'conv_act'):
            x, w = put(a['x'], a['sx']), put(a['w'], 'f')
            b = put(a['b'].reshape(-1), 'f') if kind == 'conv_act' else None
            y = ops.conv2d(x, w, b, act=('leaky' if a.get('act') else None), wscale=a['ws'])
            self.same(kind, y, out, a['so'], "%s -> %d" % (tuple(x.shape), w.shape[3]))
            if kind == 'conv_act' and a['sx'] == 'b' and ops.conv2d_pixelnorm_takes(x, w):
                # where the product runs the layer as conv + pixel norm from ONE kernel: both outputs against the emulation
                from oracle import gan_bf16_ref as emu
                y1, yn = ops.conv2d_pixelnorm(x, w, b, act=('leaky' if a.get('act') else None), wscale=a['ws'], eps=1e-8)
                self.same('conv_act+pixelnorm (fused)', y1, out, 'b', "y %s" % (tuple(x.shape),))
                # (the norm of the kernel's OWN stored y: a y value one ulp off the emulation's moves its pixel's factor)
                self.same('conv_act+pixelnorm (fused)', yn, emu.q(emu._pn(y1.detach().float().cpu().double(), 1e-8)), 'b',
                          "ynorm %s" % (tuple(x.shape),))
        elif kind == 'dgrad':
            dy, w = put(a['dy'], a['so']), put(a['w'], 'f')
            self.same(kind, ops.conv_dgrad_raw(dy, w, a['ws']), out, a['sx'], "%s -> %d" % (tuple(dy.shape), w.shape[2]))
        elif kind == 'wgrad':
            x, dy = put(a['x'], a['sx']), put(a['dy'], a['so'])
            dw, _ = ops.conv_wgrad_raw(x, dy, a['K'], want_bias=False, dw_scale=a['ws'])
            self.same(kind, dw, out, 'f', "%s x %s" % (tuple(x.shape), tuple(dy.shape)))
        elif kind == 'bias_grad':
            x, dy = put(a['x'], a['sx']), put(a['dpre'], a['so'])
            _, db = ops.conv_wgrad_raw(x, dy, a['K'], want_bias=True)
            self.same(kind, db, out, 'f', "%s" % (tuple(dy.shape),))
        elif kind == 'act_bwd':
            self.same(kind, ops.act_bwd(put(a['dy'], a['s']), put(a['y'], a['s']), 'leaky'), out, a['s'], tuple(out.shape))
        elif kind == 'pixelnorm':
            self.same(kind, ops.pixelnorm(put(a['x'], a['s']), a['eps']), out, a['s'], tuple(out.shape))
        elif kind == 'pixelnorm_bwd':
            self.same(kind, ops.pixelnorm_bwd(put(a['x'], a['s']), put(a['g'], a['s']), a['eps']), out, a['s'], tuple(out.shape))
        elif kind == 'pixelnorm_bwd2':
            dg, dx2 = ops.pixelnorm_bwd2(put(a['x'], a['s']), put(a['g'], a['s']), put(a['v'], a['s']), a['eps'])
            self.same(kind, dx2, out[0], a['s'], "dx2 %s" % (tuple(out[0].shape),))
            self.same(kind, dg, out[1], a['s'], "dg %s" % (tuple(out[1].shape),))
        elif kind == 'pool':
            x = put(a['x'], a['s'])
            y = ops.avgpool2x2(x) if (a['scale'] == 0.25 and x.shape[-1] % 4 == 0) else ops.sumpool2x2(x, a['scale'])
            self.same(kind, y, out, a['s'], tuple(x.shape))
        elif kind == 'bcast':
            self.same(kind, ops.broadcast2x2(put(a['x'], a['s']), a['scale']), out, a['s'], tuple(out.shape))
        elif kind == 'cast':
            self.same(kind, ops.cast(put(a['x'], a['s_from']), BF if a['s_to'] == 'b' else torch.float32), out, a['s_to'],
                      tuple(out.shape))
        elif kind == 'grad_sum':                                # the framework's add of bf16 gradients: nothing of ours to replay
            self.count[kind] = self.count.get(kind, 0) + 1
        else:
            raise AssertionError("unknown leaf %r" % kind)


def test_level6_every_operator_instance_of_the_step_against_the_emulation():
    """Config 5's own graph (7 levels, filters 512 .. 8, 256x256 images, penalty active), batch 2: every convolution, dgrad,
    weight gradient, bias gradient, activation backward, pixel norm (three orders), pool, broadcast and cast the
    discriminator and generator steps evaluate -- about 470 operator instances over the forward pass, the penalty's
    create_graph backward and the second-order pass -- replayed through the HIP operators on the emulation's operands.
    This is the level-6 pin of the kernels at the shapes and dispatch paths (mosaics, split reductions, ragged 8-channel
    forms, dense kernels) the real step uses; the wiring of the tape is shape-independent code pinned by the whole-graph
    test above at levels 0 - 2."""
    from tests.test_gpu_gan import make_gan
    from oracle import gan_bf16_ref as emu
    level, nb = 6, 2
    g = make_gan(dtype="bf16", num_levels=7, batch_size=nb)
    rng = np.random.default_rng(4)
    t64 = lambda a: torch.as_tensor(a, dtype=torch.float64)
    x, z, r = t64(rng.standard_normal((nb, 256, 256, 2))), t64(rng.standard_normal((nb, 1, 1, 512))), t64(rng.random(nb))
    sd = g.store.state_dict()
    det = {}
    W = emu.to_torch(sd)
    probe = {}                                                  # make the penalty active: scale the last layer
    emu.losses(x, z, 1.0, r, W, g.filters, level, details=probe)
    W["GAN/discriminator/output/logits/kernel"] = (W["GAN/discriminator/output/logits/kernel"].detach()
                                                   * (3.0 / float(probe["grad_norm"].min()))).requires_grad_(True)
    d_vars, g_vars = g.get_training_variables(level)
    rp = _Replay()
    emu.OBSERVER = rp
    try:
        with g.precision():
            _, ed, eg = emu.losses(x, z, 1.0, r, W, g.filters, level, details=det)
            torch.autograd.grad(ed, [W[n] for n, _ in d_vars], retain_graph=True, allow_unused=True)
            torch.autograd.grad(eg, [W[n] for n, _ in g_vars], allow_unused=True)
    finally:
        emu.OBSERVER = None
    assert float(det["grad_norm"].min()) > 1.05
    print("level 6 operator instances replayed:", dict(sorted(rp.count.items())))
    print("worst per kind:", {k: float("%.4g" % v) for k, v in sorted(rp.worst.items())})
    c = rp.count
    assert c["conv_act"] >= 3 * 14 + 15 and c["dgrad"] >= 60 and c["wgrad"] >= 60 and c["conv"] >= 14
    assert c["pixelnorm_bwd2"] == 2 and c["pixelnorm"] >= 17 and c["pool"] >= 18 and c["bcast"] >= 18 and c["act_bwd"] >= 60


def test_level6_whole_graph_gap_is_the_bf16_noise_floor():
    """Level 6, whole graph, batch 4.  Between ANY two implementations that store bf16 and do not add in the same order,
    stored values start to differ by one ulp where an f32 sum lands across a rounding boundary (0.04 % of the first
    layer's values); a layer whose inputs differ in a fraction p of their values stores outputs that differ in about
    0.75 sqrt(p) of theirs, so after five layers more than half of the stored values differ by an ulp and the two
    forward passes are as far apart as either is from fp64 (profiles/r04_gan_emulation_divergence.txt: 1.6e-5, 2e-4,
    7.7e-4, 1.7e-3 ... 1.4e-2 over the generator's 13 convolutions; exact-equal fraction 0.9996 -> 0.25).  So at 26+
    stored layers a whole-graph comparison with ANOTHER implementation cannot be tighter than the roundings themselves;
    what this test asserts is exactly that: HIP sits as close to the emulation as the emulation sits to fp64 (the same
    noise, not more), tape == solver path, and the first-order (generator) gradients, which are well conditioned,
    inside 0.25.  The absolute pins at level 6 are the operator replay above."""
    res = _whole_graph(6, 1.0, 7, 4)
    names = res["names"]
    r_, t_, s_ = np.array(res["rounding"]), np.array(res["tape"]), np.array(res["solver"])
    assert np.mean(t_) <= 1.25 * np.mean(r_) and np.mean(s_) <= 1.25 * np.mean(r_), (np.mean(r_), np.mean(t_), np.mean(s_))
    gen = np.array([n.startswith("GAN/generator/") for n in names])
    assert t_[gen].max() <= 0.25 and s_[gen].max() <= 0.25, (t_[gen].max(), s_[gen].max())
    assert np.abs(t_ - s_).max() <= 5e-3                         # the solver path and the plain tape: the same numbers
    _, e, t, so = res["g"]
    assert abs(t - e) <= 0.02 * max(1.0, abs(e)) and abs(so - e) <= 0.02 * max(1.0, abs(e))


def test_storage_boundaries_and_dtypes():
    """where bf16 starts and ends: generator latent block and discriminator output block are f32, every feature map between
    them bf16, images / logits f32; no framework cast kernel is needed (the autograd engine would insert one silently if
    a backward returned the wrong dtype -- checked here on the gradient dtypes of the leaves that can see it)"""
    from tests.test_gpu_gan import make_gan, dev
    g = make_gan(dtype="bf16")
    g.set_level(2)
    rng = np.random.default_rng(0)
    z = dev(rng.standard_normal((4, 1, 1, 512)).astype(np.float32))
    x = dev(rng.standard_normal((4, 16, 16, 2)).astype(np.float32)).requires_grad_(True)
    with g.precision():
        outs, last = g.generator(z, g.filters[:3])
        layers, logits = g.discriminator(x, g.filters[:3][::-1])
        assert all(o.dtype == torch.float32 for o in outs) and logits.dtype == torch.float32
        assert all(t.dtype == BF for t in layers)
        (gx,) = torch.autograd.grad(logits.sum(), x, create_graph=True)
        assert gx.dtype == torch.float32
        pen = F.dot_per_sample(gx, gx).sum()
        grads = torch.autograd.grad(pen, [v for _, v in g.discriminator_training_variables(2)], allow_unused=True)
    assert all(t is None or (t.dtype == torch.float32 and torch.isfinite(t).all()) for t in grads)
    with torch.no_grad():                                       # outside the precision context the same weights run in f32
        _, logits32 = g.discriminator(x.detach(), g.filters[:3][::-1])
    assert logits32.dtype == torch.float32
    assert np.allclose(logits.detach().cpu().numpy(), logits32.cpu().numpy(), rtol=0.05, atol=0.05)
    img = g.predict(latent=np.ones((2, 1, 1, 512), np.float32))
    assert img.dtype == torch.float32 and tuple(img.shape) == (2, 16, 16, 2)


def test_grouped_weight_gradients_mosaic_ragged_and_accumulating():
    """ops_bf16.WgradQueue on the GAN's shapes: small-image batches as mosaics, 8-channel (ragged) layers, the equalised-LR
    factor, and several contributions to ONE destination (first writes, later ones accumulate, in separate launches) --
    against the single launches of ops.conv_wgrad_raw summed up by hand."""
    from sequitr_amd import ops_bf16 as ob
    rng = np.random.default_rng(5)
    cases = [(6, 4, 4, 32, 32, 3), (5, 8, 8, 64, 32, 3), (2, 32, 32, 8, 16, 3), (2, 32, 32, 16, 8, 3), (2, 32, 32, 32, 32, 3),
             (2, 16, 16, 64, 64, 3)]
    items, want = [], []
    for (N, H, W, Cin, Cout, K) in cases:
        ws = float(np.sqrt(np.float32(2.0 / (K * K * Cout))))
        contrib = []
        for rep_ in range(3 if Cin == 32 and H == 32 else 2):     # 2 - 3 passes contribute to every weight
            xg, _ = rb(rng, (N, H, W, Cin))
            dg, _ = rb(rng, (N, H, W, Cout))
            contrib.append((xg, dg))
        dw = torch.full((K, K, Cin, Cout), 7.0, device="cuda")    # stale contents: the first contribution must overwrite them
        db = torch.full((Cout,), 7.0, device="cuda")
        items.append((contrib, K, ws, dw, db, ops._mosaic_plan(N, H, W) if W < 16 else None))
        singles = [ops.conv_wgrad_raw(a, b, K, want_bias=True, dw_scale=ws) for a, b in contrib]
        want.append((sum(s[0].double() for s in singles), sum(s[1].double() for s in singles)))
    with ob.deferred_wgrads() as q:
        for pass_ in range(3):                                   # interleaved, as a backward pass meets them
            for contrib, K, ws, dw, db, plan in items:
                if pass_ < len(contrib):
                    q.push(contrib[pass_][0], contrib[pass_][1], K, dw, db if pass_ != 1 else None, dw_scale=ws, mosaic=plan)
    for (contrib, K, ws, dw, db, plan), (rw, rb_) in zip(items, want):
        assert float((dw.double() - rw).abs().max()) <= 1e-5 * float(rw.abs().max()), (K, tuple(dw.shape))
    # db: passes 0 and 2 carried a bias destination (pass 1 did not): the sum of those two / that one
    for (contrib, K, ws, dw, db, plan) in items:
        ref = sum(ops.conv_wgrad_raw(a, b, K, want_bias=True)[1].double() for i, (a, b) in enumerate(contrib) if i != 1)
        assert float((db.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-6


def test_bf16_storage_trains_like_the_f32_graph():
    """Twelve D + G solver steps at level 2 (16 x 16) from the same initial weights, data and mixing draws in f32, mixed and
    bf16 storage: the loss trajectories stay together (the first steps to the rounding of one evaluation, later ones drift as
    the weights do), nothing blows up, and the weights move by comparable amounts."""
    from tests.test_gpu_gan import make_gan, dev
    runs = {}
    for dtype in ("f32", "mixed", "bf16"):
        g = make_gan(dtype=dtype, seed=11)
        g.set_level(2)
        w0 = {k: v.copy() for k, v in g.store.state_dict().items()}
        rng = np.random.default_rng(4)
        traj = []
        for it in range(12):
            z = dev(rng.standard_normal((4, 1, 1, 512)).astype(np.float32))
            x = dev((rng.standard_normal((4, 16, 16, 2)) * 0.5 + 0.3).astype(np.float32))
            r = dev(rng.random(4).astype(np.float32))
            g.d_solver(x, z, 1.0, r=r)
            g.g_solver(x, z, 1.0)
            traj.append(g.last_losses)
        w1 = g.store.state_dict()
        moved = np.sqrt(sum(float(((w1[k] - w0[k]) ** 2).sum()) for k in w0))
        runs[dtype] = (np.array(traj), moved)
    f32, mixed, bf16 = runs["f32"], runs["mixed"], runs["bf16"]
    assert np.isfinite(bf16[0]).all() and np.isfinite(mixed[0]).all()
    scale = 1.0 + np.abs(f32[0])
    dm, db = np.abs(mixed[0] - f32[0]) / scale, np.abs(bf16[0] - f32[0]) / scale
    assert db[0].max() <= 0.1, db[0]                            # first evaluation: one pass of rounding
    assert db.max() <= max(3.0 * dm.max(), 0.25), (dm.max(), db.max())   # later: no further from f32 than 3 x the mixed form is
    assert 0.7 <= bf16[1] / f32[1] <= 1.4 and 0.7 <= mixed[1] / f32[1] <= 1.4, (f32[1], mixed[1], bf16[1])


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 16, 16, 16), (1, 32, 48, 8, 16), (2, 64, 32, 32, 32), (1, 16, 16, 64, 64)])
def test_conv_with_average_pool_epilogue(N, H, W, Cin, Cout):
    """ops.conv2d_avgpool: the discriminator block's second conv and its 2x2 average pool from one kernel -- y and the pooled
    tensor equal the two launches bit for bit; the tape entry's gradients (first order and the penalty's second order)
    equal those of conv2d followed by avgpool2x2"""
    rng = np.random.default_rng(N + H + W + Cin + Cout)
    xg, _ = rb(rng, (N, H, W, Cin))
    w = torch.as_tensor(rng.standard_normal((3, 3, Cin, Cout)), dtype=torch.float32).cuda()
    b = torch.as_tensor(rng.standard_normal(Cout) * 0.1, dtype=torch.float32).cuda()
    ws = float(np.sqrt(np.float32(2.0 / (9 * Cout))))
    y, p = ops.conv2d_avgpool(xg, w, b, act="leaky", wscale=ws)
    y0 = ops.conv2d(xg, w, b, act="leaky", wscale=ws)
    assert torch.equal(y, y0) and torch.equal(p, ops.avgpool2x2(y0))
    # gradients: fused tape entry vs the two ops, first order and through a create_graph pass
    res = []
    for fused in (True, False):
        xa = xg.clone().requires_grad_(True)
        wa, ba = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        out = F.conv2d_avgpool(xa, wa, ba, act="leaky", wscale=ws) if fused else F.avgpool2x2(F.conv2d(xa, wa, ba, act="leaky", wscale=ws))
        gp, _ = rb(np.random.default_rng(1), tuple(out.shape))
        (gx,) = torch.autograd.grad(out, xa, gp, create_graph=True)
        pen = (gx.float() ** 2).sum()
        g1 = torch.autograd.grad(out, [xa, wa, ba], gp, retain_graph=True)
        g2 = torch.autograd.grad(pen, [wa], allow_unused=True)
        res.append((out.detach(), gx.detach(), g1, g2))
    (o1, x1, a1, s1), (o2, x2, a2, s2) = res
    assert torch.equal(o1, o2) and torch.equal(x1, x2)
    for u, v in zip(a1, a2):
        assert torch.equal(u, v)
    for u, v in zip(s1, s2):
        assert (u is None) == (v is None) and (u is None or torch.equal(u, v))


@pytest.mark.parametrize("graph", [False, True])
def test_filter_packs_follow_the_network_whose_weights_moved(graph):
    """GenerativeAdverserialNetwork._pack_filters: d_solver leaves the generator's packs valid, g_solver the discriminator's
    (one pack launch per network and iteration in the alternating loop) -- against a twin that repacks everything before every
    step: same weights, bit for bit, after D, G, D, D, G, an outside write to a generator weight, G, D; and the launch
    counts: 2 packs on the first step, then exactly one per step while the steps alternate."""
    from tests.test_gpu_gan import make_gan, dev
    rng = np.random.default_rng(21)
    zs = [dev(rng.standard_normal((4, 1, 1, 512)).astype(np.float32)) for _ in range(8)]
    xs = [dev(rng.standard_normal((4, 16, 16, 2)).astype(np.float32)) for _ in range(8)]
    rs = [dev(rng.random(4).astype(np.float32)) for _ in range(8)]
    out = []
    for lazy in (True, False):
        g = make_gan(dtype="bf16", graph=graph)
        g.set_level(2)
        g._torch_rng.manual_seed(3)
        runs = []
        real = ops.FilterPackPlan.run

        def counted(plan, _runs=runs):
            _runs.append(1)
            return real(plan)
        ops.FilterPackPlan.run = counted
        try:
            per_step = []
            for i, kind in enumerate("dgddg" + "gd"):
                if i == 5:                                      # somebody else writes a generator weight
                    name = "GAN/generator/layer_0/conv1/filter"
                    g.store.load_state_dict({name: (g.store.vars[name].detach() * 1.01).cpu().numpy()})
                if not lazy:
                    g._pack_epoch = -1                          # the twin: everything is stale before every step
                n0 = len(runs)
                (g.d_solver(xs[i], zs[i], 1.0) if kind == 'd' else g.g_solver(xs[i], zs[i], 1.0))
                per_step.append(len(runs) - n0)
        finally:
            ops.FilterPackPlan.run = real
        torch.cuda.synchronize()
        out.append(({k: v.detach().clone() for k, v in g.store.vars.items()}, per_step))
    if not graph:
        #        D  G  D  D  G | write | G  D     (D after D: its own weights moved; G after the outside write: both)
        assert out[0][1] == [2, 1, 1, 1, 1, 2, 1], out[0][1]
        assert out[1][1] == [2] * 7                             # (a capture step packs again after its recorded Adam)
    for k in out[1][0]:
        assert torch.equal(out[0][0][k], out[1][0][k]), k


@pytest.mark.parametrize("batch_d", [True, False])
def test_parameter_gradients_through_sinks_equal_the_autograd_sums(batch_d, monkeypatch):
    """GenerativeAdverserialNetwork._param_grads (dtype 'bf16'): weight gradients queued into per-parameter sinks and run as
    grouped launches (first contribution writes, later ones accumulate) against the same step with the queue switched off --
    every contribution returned through autograd and summed by the framework.  batch_d=False: D(Gz), D(X) and D(mix) are
    three separate passes, so a discriminator weight receives up to four contributions.  Same products; the sums are grouped
    differently (blocks per layer, order of the contributions): equal to f32 rounding."""
    from sequitr_amd import ops_bf16 as ob
    from tests.test_gpu_gan import make_gan, dev
    rng = np.random.default_rng(8)
    z = dev(rng.standard_normal((4, 1, 1, 512)).astype(np.float32))
    x = dev(rng.standard_normal((4, 16, 16, 2)).astype(np.float32))
    r = dev(rng.random(4).astype(np.float32))

    def grads(group):
        monkeypatch.setattr(ob, "WGRAD_GROUP_MAX_ELEMS", (1 << 31) if group else 0)
        g = make_gan(dtype="bf16", batch_d=batch_d)
        g.set_level(2)
        with g.precision(), F.fuse_act_gates(True):
            d_vars, gd, _ = g._d_grads(x, z, 0.7, r)
            g_vars, gg, _ = g._g_grads(x, z, 0.7)
        return [n for n, _ in d_vars + g_vars], [None if t is None else t.detach().clone() for t in list(gd) + list(gg)]
    names, a = grads(True)
    _, b = grads(False)
    assert sum(t is not None for t in a) == sum(t is not None for t in b) >= 20
    for n, u, v in zip(names, a, b):
        assert (u is None) == (v is None), n
        if u is not None:
            assert float((u - v).abs().max()) <= 2e-5 * float(v.abs().max()) + 1e-7, n
