import importlib, os, sys
from types import SimpleNamespace as NS
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module(bench.PKG)
nftrain = importlib.import_module(bench.PKG + '.train')
nfdata = importlib.import_module(bench.PKG + '.data')
cfg = bench.CONFIGS['c4']
dev = torch.device('cuda:0')
torch.manual_seed(0); np.random.seed(0)
net = getattr(pkg, cfg['cls'])(cfg['dims'], cfg['datatype'], NS(layers=cfg['layers'], mixtures=cfg['mixtures'])).to(dev)
trainer = nftrain.FlowTrainer(net, graph=False, warmup=2)
y = nfdata.sample(cfg['data'], cfg['batch'], 1234).reshape((cfg['batch'], ) + cfg['dims']).to(dev)
for _ in range(3):
    trainer.train_on_batch(y)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    trainer.train_on_batch(y)
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_stack_n=8)
sel = [r for r in rows if any(k in r.key.lower() for k in ('memcpy', 'copy_', 'clone', 'contiguous', 'memset', 'fill_', 'zero_'))]
sel.sort(key=lambda r: -r.count)
for r in sel[:25]:
    print('%6d x %-28s cuda %.1f us' % (r.count, r.key[:28], r.device_time_total))
    for fr in r.stack[:8]:
        print('          ', fr[:150])
