import importlib, os, sys
from types import SimpleNamespace as NS
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import trajectory as traj
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
nfdata = importlib.import_module('normalizing-flows-pytorch_amd.data')
nftrain = importlib.import_module('normalizing-flows-pytorch_amd.train')
pairs = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv), 2)]
torch.manual_seed(0); np.random.seed(0)
dims = (3, 32, 32); B = 64
net = pkg.Glow(dims, 'image', NS(layers=32, mixtures=None))
y = nfdata.sample('cifar', B, 1234).reshape((B,) + dims)
net = net.to('cuda')
trainer = nftrain.FlowTrainer(net, graph=False)
trainer.train_on_batch(y.cuda()); torch.cuda.synchronize()
sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
for a, b in pairs:
    r64 = traj.run_slice('glow', dims, 'image', 32, sd, a, b, None, None, dtype=torch.float64, y=y)
    z_in = r64['z_in'].float().cuda().requires_grad_(True); ld_in = r64['ld_in'].float().cuda()
    net.train()
    def fl():
        z, ld = net.forward_slice(z_in, ld_in.clone(), a, b)
        return z, nftrain.nll_loss(z, ld)
    z, loss = trainer._run_step(z_in.device, fl)
    g = z_in.grad.detach().double().cpu(); w = r64['g_in']
    err = (g - w).abs().reshape(B, -1).max(1).values / w.abs().max()
    print('slice', a, b, 'z err %.2e' % float((z.detach().cpu().double() - r64['z']).abs().max()), 'g_in rel err max %.2e' % float(err.max()), 'rows bad', int((err > 1e-4).sum()))
    e = (g - w).abs() / w.abs().max()
    print('per channel max err:', ['%.1e' % float(e[:, c].max()) for c in range(0, e.shape[1], max(1, e.shape[1] // 12))])
    bad = {}
    for k, want in r64['grads'].items():
        p = dict(net.named_parameters())[k]
        er = float((p.grad.detach().cpu().double() - want).abs().max() / max(1.0, float(want.abs().max())))
        if er > 1e-4: bad[k] = er
    print('bad param grads', list(bad.items())[:8])
