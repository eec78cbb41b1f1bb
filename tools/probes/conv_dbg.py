import copy, importlib, sys
import torch
sys.path.insert(0, '.')
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
cond = importlib.import_module('normalizing-flows-pytorch_amd.conditioners')
I, O, H, W, B, training = [int(v) for v in sys.argv[1:7]]
torch.manual_seed(I * 100 + O)
a = cond.ConvNet(I, O).cuda()
with torch.no_grad():
    for m in a.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.3)
b = copy.deepcopy(a); b.fused = False
a.train(bool(training)); b.train(bool(training))
x1 = torch.randn(B, I, H, W, device='cuda').requires_grad_(True)
x2 = x1.detach().clone().requires_grad_(True)
y1, y2 = a(x1), b(x2)
w = torch.randn_like(y2)
(y1 * w).sum().backward(); (y2 * w).sum().backward()
err = (x1.grad - x2.grad).abs()
tol = 2e-4 * max(1.0, float(x2.grad.abs().max()))
bad = err > tol
print('bad frac', float(bad.float().mean()), 'max', float(err.max()))
print('per sample', bad.sum((1, 2, 3)).tolist())
print('per channel', bad.sum((0, 2, 3)).tolist())
print('per y', bad.sum((0, 1, 3)).tolist())
print('per x', bad.sum((0, 1, 2)).tolist())
pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
for n, p in pb.items():
    e = float((pa[n].grad - p.grad).abs().max()); m = float(p.grad.abs().max())
    print('%-40s err %.2e max %.2e' % (n, e, m))
