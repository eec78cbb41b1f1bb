"""device time of the C1 whole-flow launches (HIP events around the C-ABI calls), current library:  python tools/probes/solo_time.py [B] [S]"""
import importlib, sys
from types import SimpleNamespace as NS
import torch
sys.path.insert(0, '.')
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
train = importlib.import_module('normalizing-flows-pytorch_amd.train')
N = pkg._native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
torch.manual_seed(0)
net = pkg.RealNVP((2, ), 'density', NS(layers=S, mixtures=8)).to('cuda').train()
tr = train.FlowTrainer(net, graph=False)
y = (torch.randn(B, 2) * 0.7).to('cuda')
for _ in range(3):
    tr._forward_backward(y)
for name in ('nf_realnvp_flow_vec_fwd', 'nf_realnvp_flow_vec_bwd_deferred'):
    with N.timed_launches(name) as tl:
        for _ in range(20):
            tr._forward_backward(y)
        d = sorted(tl.durations_us())
    print('%-36s median %.1f us  min %.1f  (%.2f us per flow step)' % (name, d[len(d) // 2], d[0], d[len(d) // 2] / S))
print('persistent timeouts', N.persistent_timeouts())
