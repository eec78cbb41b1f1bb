"""Where does a member's dominant ReLU-decision event sit?  C1, initial weights (the state of the parity test's step 1): for every GPU / fp32-oracle
member (row permutation) print distance@origin and, for the origin step, the three gradient tensors with the largest error relative to their own
largest entry and the unit (row of the weight) that carries it.  An event of ONE unit of layer j shows as one row of layer j's weight gradient
(and one entry of its bias / BatchNorm gradients); an event of the 1 -> 32 first layer (all units decide on sign(x - mean)) shows in all rows.
    python tools/probes/kink_locate.py [N]"""
import importlib, os, sys
from types import SimpleNamespace as NS
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from oracle import trajectory as traj  # noqa: E402
import parity_modes as pm              # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
nfdata = importlib.import_module('normalizing-flows-pytorch_amd.data')
nftrain = importlib.import_module('normalizing-flows-pytorch_amd.train')
torch.set_num_threads(8)
torch.manual_seed(0); np.random.seed(0)
net = pkg.RealNVP((2, ), '2d', NS(layers=32, mixtures=None))
y = nfdata.sample('moons', 256, 1234)
sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
net = net.to('cuda')
tr = nftrain.FlowTrainer(net, graph=False)
yd = y.to('cuda')
r64 = traj.run('realnvp', (2, ), '2d', 32, sd, y, 1, dtype=torch.float64)[0][1]
gp = torch.Generator().manual_seed(99)
perms = [torch.arange(256)] + [torch.randperm(256, generator=gp) for _ in range(N - 1)]


def describe(g, o):
    rows = []
    for k, e in r64['grads'].items():
        if k in g and k.startswith('net.layers.') and int(k.split('.')[2]) // 2 == o and e.dim() >= 2 and e.numel() > 32:
            d = (g[k].double() - e.double()).abs()
            rel = float(d.max()) / max(1e-30, float(e.abs().max()))
            idx = int(d.reshape(d.shape[0], -1).max(1).values.argmax()) if d.dim() >= 1 and d.numel() > 1 else 0
            # how concentrated: share of the squared error in the worst row
            rowsq = (d.reshape(d.shape[0], -1) ** 2).sum(1) if d.dim() >= 1 and d.numel() > 1 else d.reshape(1) ** 2
            conc = float(rowsq.max() / max(1e-300, float(rowsq.sum())))
            rows.append((rel, k.replace('net.layers.', 'L'), idx, conc))
    rows.sort(reverse=True)
    return '  '.join('%s %.1e unit %d (%.0f%% of the error in that row)' % (k, r, i, 100 * c) for r, k, i, c in rows[:4])


for who in ('gpu', 'oracle'):
    print('==', who)
    for n, p in enumerate(perms):
        if who == 'gpu':
            net.load_state_dict(sd)
            tr._forward_backward(yd[p.to('cuda')])
            torch.cuda.synchronize()
            g = {k: q.grad.detach().cpu().clone() for k, q in net.named_parameters() if q.grad is not None}
        else:
            g = traj.run('realnvp', (2, ), '2d', 32, sd, y[p], 1, dtype=torch.float32)[0][1]['grads']
        prof = pm.step_profile(g, r64, 2, 32)
        o = pm.origin(prof)
        # the event step: the LAST step whose weight-matrix gradients are off by more than 10 x the median step error of the later steps
        ev = o
        print('  member %2d  %.2e @ step %2d : %s' % (n, pm.flat(g, r64), o, describe(g, o)), flush=True)
        print('             profile (worst entry / max entry of the multi-element tensors, steps 0 .. 31): ' + ' '.join('%.0e' % v for v in prof), flush=True)
