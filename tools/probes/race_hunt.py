"""repeat the persistent-kernel paths at several grid sizes and report run-to-run differences (race hunting)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
fused = importlib.import_module(pkg.__name__ + '.fused')
cond = importlib.import_module(pkg.__name__ + '.conditioners')
dev = torch.device('cuda:0')
torch.manual_seed(0)
mlp = cond.MLP(1, 2).to(dev).train()
a, c, k = pkg.ActNorm((2, )), pkg.InvertibleConv1x1(2), pkg.AffineCoupling((2, ))
mods = torch.nn.ModuleList([a, c, k]).to(dev).train()
a.initialized = True
for n in (4096, 8192, 12288, 16384):
    x = torch.randn(n, 1, device=dev)
    z = torch.randn(n, 2, device=dev)
    gy = torch.randn(n, 2, device=dev)
    res = {'mlp': [], 'glow': []}
    for it in range(12):
        xx = x.clone().requires_grad_(True)
        y = fused.mlp_forward(mlp, xx, chain=True)
        y.backward(gy)
        res['mlp'].append((y.detach().clone(), xx.grad.clone()))
        zz = z.clone().requires_grad_(True)
        yy, ld = fused.glow_step_vec(zz, torch.zeros(n, device=dev), a, c, k)
        (yy * gy).sum().backward()
        res['glow'].append((yy.detach().clone(), zz.grad.clone()))
        for p in list(mlp.parameters()) + list(mods.parameters()):
            p.grad = None
    for key, lst in res.items():
        dy = max(float((t[0] - lst[0][0]).abs().max()) for t in lst)
        dg = max(float((t[1] - lst[0][1]).abs().max()) for t in lst)
        print('N %6d %-5s max run-to-run diff: out %.3e  input-grad %.3e' % (n, key, dy, dg))
