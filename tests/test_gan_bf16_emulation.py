"""CPU: the bf16 rounding-point emulation of the GAN (oracle/gan_bf16_ref.py) -- the checker of the -m gpu tests in
tests/test_gpu_gan_bf16.py -- pinned to the fp64 restatement oracle/torch_gan_ref.py (gan.py:149-316, 665-732):
with its roundings switched off the emulation's closed operator set (conv / dgrad / wgrad as each other's
derivatives, pixel norm to second order, pool <-> broadcast, the bf16 gradient forks) must reproduce plain autograd
on the fp64 graph, losses and every parameter gradient including the penalty's second-order terms."""
import numpy as np
import pytest
import torch

from oracle import gan_bf16_ref as emu
from oracle import torch_gan_ref as ref

FILTERS = [32, 16, 8]


def weights(level, seed=0, bias_std=0.1):
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape in emu.variable_shapes(FILTERS, level).items():
        if name.endswith('/bias'):
            w[name] = rng.standard_normal(shape) * bias_std
        elif name.endswith('/kernel'):
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            w[name] = rng.uniform(-lim, lim, shape)
        else:
            w[name] = rng.standard_normal(shape)
    return w


def inputs(level, n=3, seed=1):
    rng = np.random.default_rng(seed)
    side = 4 * 2 ** level
    t = lambda a: torch.as_tensor(a, dtype=torch.float64)
    return t(rng.standard_normal((n, side, side, 2))), t(rng.standard_normal((n, 1, 1, 512))), t(rng.random(n))


def grads(mod, W, X, Z, r, level, alpha):
    _, d_loss, g_loss = mod.losses(X, Z, alpha, r, W, FILTERS, level)
    dn = [k for k in W if k.startswith('GAN/discriminator/')]
    gn = [k for k in W if k.startswith('GAN/generator/')]
    dg = torch.autograd.grad(d_loss, [W[k] for k in dn], retain_graph=True, allow_unused=True)
    gg = torch.autograd.grad(g_loss, [W[k] for k in gn], allow_unused=True)
    return d_loss.item(), g_loss.item(), dict(zip(dn + gn, dg + gg))


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


@pytest.mark.parametrize("level,alpha", [(0, 1.0), (1, 0.3), (2, 0.7)])
def test_emulation_without_rounding_is_the_fp64_restatement(level, alpha):
    w = weights(level)
    X, Z, r = inputs(level)
    if level == 0:                                  # torch_gan_ref evaluates every to_image: give it the ones the losses skip
        pass
    W1, W2 = emu.to_torch(w), ref.to_torch(w)
    with emu.rounding(False):
        d1, g1, G1 = grads(emu, W1, X, Z, r, level, alpha)
    # torch_gan_ref builds to_image at EVERY level: add the variables the losses never read (no gradient)
    for l in range(level + 1):
        for v, shape in (('filter', (1, 1, FILTERS[l], 2)), ('bias', (1, 1, 1, 2))):
            k = 'GAN/generator/to_image/to_image%d/%s' % (l, v)
            if k not in W2:
                W2[k] = torch.zeros(shape, dtype=torch.float64, requires_grad=True)
    d2, g2, G2 = grads(ref, W2, X, Z, r, level, alpha)
    assert abs(d1 - d2) <= 1e-10 * max(1.0, abs(d2)) and abs(g1 - g2) <= 1e-10 * max(1.0, abs(g2))
    for k, g in G1.items():
        assert (g is None) == (G2[k] is None), k
        if g is not None:
            assert rel(g, G2[k]) <= 1e-9, (k, rel(g, G2[k]))
    # the generator step's own evaluation (through the discriminator alone) gives the same generator gradients
    with emu.rounding(False):
        gl = emu.generator_loss(X, Z, alpha, W1, FILTERS, level)
        gn = [k for k in W1 if k.startswith('GAN/generator/')]
        for k, g in zip(gn, torch.autograd.grad(gl, [W1[k] for k in gn], allow_unused=True)):
            assert rel(g, G2[k]) <= 1e-9, k


def test_rounding_points_store_bf16_values_and_move_the_gradients():
    """with the roundings on: every stored feature value is a bfloat16 value, the losses move by about 2^-9 relative
    and the parameter gradients by the amounts the -m gpu tests print beside the HIP-vs-emulation gap"""
    level, alpha = 2, 1.0
    w = weights(level)
    X, Z, r = inputs(level)
    W = emu.to_torch(w)
    imgs = emu.generator(Z, W, FILTERS[:level + 1])
    assert sorted(imgs) == [1, 2] and imgs[2].shape == (3, 16, 16, 2)
    t = emu.q(torch.as_tensor([1.0 + 2.0 ** -9, 1.0 + 3 * 2.0 ** -9, -0.3], dtype=torch.float64))
    assert t[0].item() == 1.0 and t[1].item() == 1.0 + 2.0 ** -7 and abs(t[2].item() + 0.3) < 2.0 ** -10     # RNE, ties to even
    feat = emu.wconv(emu.q(torch.randn(2, 8, 8, 16, dtype=torch.float64)), W, 'GAN/discriminator/layer_1/conv1', 'b', 'b', norm=False)
    assert torch.equal(feat, emu.q(feat))                                                       # stored values ARE bf16 values
    d1, g1, G1 = grads(emu, W, X, Z, r, level, alpha)
    with emu.rounding(False):
        d0, g0, G0 = grads(emu, emu.to_torch(w), X, Z, r, level, alpha)
    assert 0 < abs(d1 - d0) <= 0.05 * max(1.0, abs(d0)) and 0 < abs(g1 - g0) <= 0.05 * max(1.0, abs(g0))
    errs = {k: rel(G1[k], G0[k]) for k in G1 if G1[k] is not None}
    print("rounding alone, level 2: mean %.4f max %.4f (%s)" % (np.mean(list(errs.values())), max(errs.values()),
                                                                max(errs, key=errs.get)))
    assert 1e-4 < np.mean(list(errs.values())) < 0.5


def test_conv_policy_restates_the_dispatch():
    P = emu.conv_policy
    assert P('conv', 3, 64, 64, 4096, 'b', 'b') == (False, True, 'b')          # feature conv: packed bf16 filter, bf16 out
    assert P('conv', 1, 2, 8, 4096, 'f', 'b') == (False, False, 'b')           # from_image: f32 multiply, stored bf16
    assert P('conv', 1, 8, 2, 4096, 'b', 'f') == (False, False, 'f')           # to_image
    assert P('conv', 1, 512, 8192, 32, 'f', 'f') == (True, True, 'f')          # generator dense1: mixed
    assert P('conv', 1, 8208, 512, 32, 'f', 'f') == (False, False, 'f')        # discriminator dense forward: f32 split reduction
    assert P('conv', 1, 512, 8208, 32, 'f', 'f') == (True, True, 'f')          # ... its dgrad: mixed
    assert P('conv', 1, 512, 1, 32, 'f', 'f') == (False, False, 'f')           # logits
    assert P('wgrad', 1, 8208, 512, 32, 'f', 'f') == (False, False, 'f')       # dense weight gradients: f32
    assert P('wgrad', 1, 528, 32, 4, 'f', 'f') == (True, False, 'f')           # the small test net's dense: mixed wgrad
    assert P('wgrad', 3, 64, 64, 4096, 'b', 'b') == (False, False, 'f')
