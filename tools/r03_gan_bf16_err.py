"""prints loss / gradient errors of the GAN's 'mixed' and 'bf16' (storage) forms against the fp64 reference"""
import numpy as np, torch, sys
sys.path.insert(0, '.')
from oracle import torch_gan_ref as ref
from tests.test_gpu_gan import make_gan, dev
for level, alpha in ((0, 1.0), (2, 0.4), (2, 1.0)):
    rng = np.random.default_rng(2)
    z = rng.standard_normal((4, 1, 1, 512)).astype(np.float32)
    r = rng.random(4).astype(np.float32)
    x = None
    res = {}
    for dtype in ("f32", "mixed", "bf16"):
        g = make_gan(dtype=dtype)
        g.set_level(level)
        if x is None:
            x = rng.standard_normal((4,) + g.get_size(level) + (2,)).astype(np.float32)
        with g.precision():
            _, d_loss, g_loss = g._build_network(dev(x), dev(z), alpha, r=dev(r))
            d_vars, g_vars = g.get_training_variables(level)
            dg = torch.autograd.grad(d_loss, [v for _, v in d_vars], retain_graph=True, allow_unused=True)
            gg = torch.autograd.grad(g_loss, [v for _, v in g_vars], allow_unused=True)
        res[dtype] = (d_loss.item(), g_loss.item(), [t.cpu().numpy() for t in dg + gg])
    W = ref.to_torch(g.store.state_dict())
    _, rd, rg = ref.losses(torch.as_tensor(x, dtype=torch.float64), torch.as_tensor(z, dtype=torch.float64), alpha,
                           torch.as_tensor(r, dtype=torch.float64), W, g.filters, level)
    rdg = torch.autograd.grad(rd, [W[n] for n, _ in d_vars], retain_graph=True, allow_unused=True)
    rgg = torch.autograd.grad(rg, [W[n] for n, _ in g_vars], allow_unused=True)
    truth = [t.numpy() for t in rdg + rgg]
    names = [n for n, _ in d_vars + g_vars]
    print("level", level, "alpha", alpha, "fp64 d/g", rd.item(), rg.item())
    for k in res:
        e = [float(np.linalg.norm((a.astype(np.float64) - t).ravel()) / max(np.linalg.norm(t.ravel()), 1e-30)) for a, t in zip(res[k][2], truth)]
        print("  %-6s d_loss %.5f g_loss %.5f  grad rel err: mean %.4f max %.4f (%s)" % (k, res[k][0], res[k][1], np.mean(e), np.max(e), names[int(np.argmax(e))]))
