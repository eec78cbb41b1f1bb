"""which framework ops (copies, fills, adds ...) still run inside one eager training step, with their call sites:
   python tools/op_count.py [c2]"""
import importlib
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = importlib.import_module(bench.PKG)
nftrain = importlib.import_module(bench.PKG + '.train')
nfdata = importlib.import_module(bench.PKG + '.data')
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'c2']
dev = torch.device('cuda:0')
torch.manual_seed(0)
np.random.seed(0)
net = getattr(pkg, cfg['cls'])(cfg['dims'], cfg['datatype'], NS(layers=cfg['layers'], mixtures=cfg['mixtures'])).to(dev)
trainer = nftrain.FlowTrainer(net, graph=False, warmup=2)
y = nfdata.sample(cfg['data'], cfg['batch'], 1234)
if cfg['data'] == 'cifar':
    y = y.reshape((cfg['batch'], ) + cfg['dims'])
y = y.to(dev)
for _ in range(3):
    trainer.train_on_batch(y)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    trainer.train_on_batch(y)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by='self_cuda_time_total', row_limit=40, max_name_column_width=50,
                                                  max_src_column_width=90))
