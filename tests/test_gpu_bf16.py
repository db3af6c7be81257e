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
         (1, 16, 16, 64, 32, 1, None), (1, 24, 24, 16, 64, 1, "relu"), (1, 8, 8, 256, 256, 3, "relu")]


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


def test_first_conv_bf16():
    x, w, b = tiles(6, 2, 40, 24, 1), rand_weights(7, (3, 3, 1, 16), 0.5), rand_weights(8, (16,), 0.1)
    ref = TF.relu(TF.conv2d(torch.as_tensor(x, dtype=torch.float64).permute(0, 3, 1, 2),
                            torch.as_tensor(w, dtype=torch.float64).permute(3, 2, 0, 1),
                            torch.as_tensor(b, dtype=torch.float64), padding=1)).permute(0, 2, 3, 1)
    check_bf16(ob.conv3x3_first(dev(x), dev(w), dev(b)), ref, "first conv")


@pytest.mark.parametrize("N,H,W,Cin,Cout,K", [(2, 32, 48, 16, 16, 3), (1, 32, 32, 32, 64, 3), (2, 21, 19, 16, 32, 3),
                                              (1, 16, 16, 64, 256, 1), (1, 16, 16, 128, 128, 3)])
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
