"""per-parameter gradients of a 2-D RealNVP train step: the one-workgroup kernels (csrc/flow_solo.hip) against the grid kernels
(csrc/mlp_chain.hip) and against the float64 oracle.   python tools/probes/solo_dbg.py [B] [layers]"""
import importlib, os, sys
from types import SimpleNamespace as NS
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
N = pkg._native
N.load()
from oracle import models as om
from oracle import transforms as tf
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.manual_seed(int(sys.argv[3]) if len(sys.argv) > 3 else 2256)
net = pkg.RealNVP((2, ), 'density', NS(layers=K, mixtures=8))
sd = om.clone_state(net.state_dict())
y = torch.randn(B, 2) * 0.7


train = importlib.import_module('normalizing-flows-pytorch_amd.train')


def run(on):
    N.call('nf_flow_solo_config', on)
    m = pkg.RealNVP((2, ), 'density', NS(layers=K, mixtures=8))
    m.load_state_dict(sd)
    m = m.to('cuda').train()
    tr = train.FlowTrainer(m, graph=False)
    yy = y.to('cuda').requires_grad_(True)
    z, loss = tr._forward_backward(yy)
    torch.cuda.synchronize()
    g = {k: v.grad.detach().cpu().clone() for k, v in m.named_parameters() if v.grad is not None}
    g['INPUT'] = yy.grad.detach().cpu().clone() if yy.grad is not None else torch.zeros(1)
    return z.detach().cpu(), torch.as_tensor(float(loss.detach())), g


z1, l1, g1 = run(3)
z0, l0, g0 = run(0)
sd64 = {k: (v.double().clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
ora = om.FlowOracle('realnvp', (2, ), 'density', K, sd64, training=True).requires_grad_(True)
zo, lo = ora.forward(y.double())
tf.nll_loss(zo, lo).backward()
ref = {k: v.grad for k, v in ora.parameters().items() if v.grad is not None}
print('z: solo vs f64 %.2e  grid vs f64 %.2e' % (float((z1.double() - zo.detach()).abs().max()), float((z0.double() - zo.detach()).abs().max())))
print('z solo vs grid %.2e   ld %.2e' % (float((z1 - z0).abs().max()), float((l1 - l0).abs().max())))
worst = []
for k in g0:
    s = max(1e-12, float(g0[k].abs().max()))
    e10 = float((g1[k] - g0[k]).abs().max())
    row = [k, s, e10 / s]
    if ref is not None and k in ref:
        row += [float((g1[k].double() - ref[k]).abs().max()) / s, float((g0[k].double() - ref[k]).abs().max()) / s]
    worst.append(row)
worst = [r for r in worst if r[1] > 1e-4]
worst.sort(key=lambda r: -r[2])
for r in worst[:25]:
    print('%-46s scale %.3e  solo-grid %.2e' % (r[0], r[1], r[2]) + ('  solo-f64 %.2e  grid-f64 %.2e' % (r[3], r[4]) if len(r) > 3 else ''))
print('--- the last flow step (first in the backward), in module order')
for k in g0:
    if k.startswith('net.layers.%d.' % (2 * K - 1)) and float(g0[k].abs().max()) > 1e-4:
        s_ = float(g0[k].abs().max())
        print('%-46s scale %.3e  solo-grid %.2e' % (k, s_, float((g1[k] - g0[k]).abs().max()) / s_))

print('--- per flow step: worst relative error of its parameters with scale > 1e-4 (solo vs grid)')
for i in range(K * 2):
    es = [float((g1[k] - g0[k]).abs().max()) / float(g0[k].abs().max()) for k in g0 if k.startswith('net.layers.%d.' % i) and float(g0[k].abs().max()) > 1e-4]
    if es:
        print('layers.%d  %.2e' % (i, max(es)))
print('input gradient: scale %.3e  solo-grid %.2e' % (float(g0['INPUT'].abs().max()), float((g1['INPUT'] - g0['INPUT']).abs().max())))
