import copy, importlib, sys
import torch
from types import SimpleNamespace as NS
sys.path.insert(0, '.')
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
train = importlib.import_module('normalizing-flows-pytorch_amd.train')
fused = importlib.import_module('normalizing-flows-pytorch_amd.fused')
B, K = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(0)
net1 = pkg.Glow((2,), 'density', NS(layers=K, mixtures=8)).cuda()
net2 = copy.deepcopy(net1)
y = (torch.randn(B, 2) * 0.7).cuda()
t1, t2 = train.FlowTrainer(net1, graph=False), train.FlowTrainer(net2, graph=False)
real = fused.glow_flow_vec_usable
for step in range(3):
    fused.glow_flow_vec_usable = real
    t1.net.train(); z1, l1 = t1._forward_backward(y)
    fused.glow_flow_vec_usable = lambda z, s: False
    t2.net.train(); z2, l2 = t2._forward_backward(y)
    err = (z1 - z2).abs().max(1).values
    bad = (err > 2e-5).nonzero().flatten()
    print('step', step, 'z err', float(err.max()), 'bad rows', bad.numel(), 'blocks', sorted(set((bad // 128).tolist()))[:20],
          'grad err', float((t1.bucket.flat - t2.bucket.flat).abs().max()), 'timeouts', fused.N.persistent_timeouts())
    # keep the two models identical for the next step
    net2.load_state_dict(net1.state_dict())
    t2.bucket.flat_params.copy_(t1.bucket.flat_params)
