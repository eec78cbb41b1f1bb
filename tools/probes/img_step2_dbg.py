"""Where does the step-2 gradient of a small image RealNVP differ from the oracle?  (tests/test_gpu_fullsize_parity.py: realnvp_24 /
realnvp_mnist step 2: flat distance to float64 1.9e-4 with the fp32 oracle at 1.3e-7, step 1 perfect.)  From the SAME state: the fused HIP
conditioners, the module-by-module conditioners (ConvNet.fused = False: ATen / MIOpen convolutions, same transforms), the oracle in float32
and float64; pairwise relative L2 distances of the flat gradient and, per flow step, the number of ReLU units of the conditioners whose
oracle pre-activation in float64 is closer to zero than 1e-5 (a unit the two sides can mask differently: a "kink event").
    python tools/probes/img_step2_dbg.py [realnvp|glow] [side] [steps_before]"""
import importlib
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
nfdata = importlib.import_module('normalizing-flows-pytorch_amd.data')
nftrain = importlib.import_module('normalizing-flows-pytorch_amd.train')
cond = importlib.import_module('normalizing-flows-pytorch_amd.conditioners')
from oracle import trajectory as traj

kind = sys.argv[1] if len(sys.argv) > 1 else 'realnvp'
side = int(sys.argv[2]) if len(sys.argv) > 2 else 24
before = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dims, B, layers = (1, side, side), 64, 2
torch.manual_seed(0)
np.random.seed(0)
net = getattr(pkg, {'realnvp': 'RealNVP', 'glow': 'Glow'}[kind])(dims, 'image', NS(layers=layers, mixtures=None))
y = nfdata.sample('cifar', B, 1234).reshape(B, -1)[:, :int(np.prod(dims))].reshape((B, ) + dims).contiguous()
net = net.cuda()
trainer = nftrain.FlowTrainer(net, graph=False)
yd = y.cuda()
for _ in range(before):
    trainer.train_on_batch(yd)
torch.cuda.synchronize()
sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}


def flat_of(rec):
    return torch.cat([rec['grads'][k].double().reshape(-1) for k in names])


def gpu_grads():
    z, loss = trainer._forward_backward(yd)
    torch.cuda.synchronize()
    return {k: p.grad.detach().double().cpu().clone() for k, p in net.named_parameters() if p.grad is not None}, float(loss)


g_fused, l_fused = gpu_grads()
g_fused2, _ = gpu_grads()
cond.ConvNet.fused = False
g_mod, l_mod = gpu_grads()
cond.ConvNet.fused = True
r32 = traj.run(kind, dims, 'image', layers, sd, y, 1, mixtures=None, dtype=torch.float32, actnorm_initialized=before > 0)[0][1]
r64 = traj.run(kind, dims, 'image', layers, sd, y, 1, mixtures=None, dtype=torch.float64, actnorm_initialized=before > 0)[0][1]
names = [k for k in r64['grads'] if k in g_fused]
f64 = flat_of(r64)


def dist(a, b):
    return float((a - b).norm() / b.norm())


F = torch.cat([g_fused[k].reshape(-1) for k in names])
F2 = torch.cat([g_fused2[k].reshape(-1) for k in names])
M = torch.cat([g_mod[k].reshape(-1) for k in names])
C32 = flat_of(r32)
print('%s (1, %d, %d), state after %d step(s): loss fused %.6f modules %.6f cpu32 %.6f' % (kind, side, side, before, l_fused, l_mod, float(r32['loss'])))
print('flat gradient, relative L2:  fused vs float64 %.3e | fused (second run) vs float64 %.3e | modules vs float64 %.3e | cpu32 vs float64 %.3e | '
      'fused vs modules %.3e | fused vs fused again %.3e' % (dist(F, f64), dist(F2, f64), dist(M, f64), dist(C32, f64), dist(F, M), dist(F, F2)))
GMAX = max(float(r64['grads'][k].double().abs().max()) for k in names)


def scale(k):      # (pre-BatchNorm biases have analytically zero gradients: bounded against the largest entry of the model)
    return max(float(r64['grads'][k].double().abs().max()), 1e-3 * GMAX)


worst = sorted(((float((g_fused[k] - r64['grads'][k].double()).abs().max()) / scale(k), k) for k in names), reverse=True)[:8]
for e, k in worst:
    em = float((g_mod[k] - r64['grads'][k].double()).abs().max()) / scale(k)
    print('   %-60s fused %.3e  modules %.3e' % (k, e, em))
