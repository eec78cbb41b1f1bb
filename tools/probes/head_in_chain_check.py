import importlib, sys, torch, numpy as np
from types import SimpleNamespace as NS
sys.path.insert(0,'/root/repo')
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
train = importlib.import_module('normalizing-flows-pytorch_amd.train')
L = importlib.import_module('normalizing-flows-pytorch_amd.layers')
N = pkg._native
torch.manual_seed(0)
net = pkg.Glow((3,32,32),'image',NS(layers=4,mixtures=None)).to('cuda')
tr = train.FlowTrainer(net, graph=False)
y = torch.rand(64,3,32,32,device='cuda')
for _ in range(2): tr.train_on_batch(y)
for on in (True, False):
    L.HEAD_IN_CHAIN = on
    with N.timed_launches('nf_glow_head_w_fwd') as t:
        z, loss = tr._forward_backward(y)
        n = len(t.durations_us())
    print('HEAD_IN_CHAIN', on, 'head fwd launches', n, 'loss', float(loss), float(z.abs().sum()))

# durations of the chain forward launches per level, with and without the head in the prologue (HIP events around every launch)
import collections
for on in (True, False):
    L.HEAD_IN_CHAIN = on
    per = collections.defaultdict(list)
    shapes = []
    with N.timed_launches('nf_convnet_chain_fwd') as t:
        orig = N.call
        for _ in range(3):
            tr._forward_backward(y)
        d = t.durations_us()
    with N.timed_launches('nf_glow_head_w_fwd') as t2:
        for _ in range(3):
            tr._forward_backward(y)
        dh = t2.durations_us()
    print('HEAD_IN_CHAIN', on, 'chain fwd launches %d, mean %.1f us, sum per step %.1f us | separate head launches %d, sum per step %.1f us'
          % (len(d) // 3, sum(d) / len(d), sum(d) / 3, len(dh) // 3, sum(dh) / 3))
