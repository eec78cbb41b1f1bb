import importlib, os, sys
from types import SimpleNamespace as NS
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module(bench.PKG)
nftrain = importlib.import_module(bench.PKG + '.train')
nfdata = importlib.import_module(bench.PKG + '.data')
NAME = sys.argv[1] if len(sys.argv) > 1 else 'c4'
cfg = bench.CONFIGS[NAME]
dev = torch.device('cuda:0')
torch.manual_seed(0); np.random.seed(0)
net = getattr(pkg, cfg['cls'])(cfg['dims'], cfg['datatype'], NS(layers=cfg['layers'], mixtures=cfg['mixtures'])).to(dev)
trainer = nftrain.FlowTrainer(net, graph=False, warmup=2)
y = nfdata.sample(cfg['data'], cfg['batch'], 1234).reshape((cfg['batch'], ) + cfg['dims']).to(dev)
GRAPH = len(sys.argv) > 2 and sys.argv[2] == 'graph'
if GRAPH:
    trainer = nftrain.FlowTrainer(net, graph=True, warmup=2)
    for _ in range(3):
        trainer.train_on_batch(y)
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

# device-side copies by the op that issued them (the runtime's copyBuffer kernel does not show up under an aten key of its own)
import collections
byop = collections.Counter()
for e in prof.events():
    n = e.name.lower()
    if 'memcpy' in n or 'copybuffer' in n:
        par = e.cpu_parent
        chain = []
        while par is not None and len(chain) < 4:
            chain.append(par.name)
            par = par.cpu_parent
        byop[(e.name[:40], ' < '.join(chain))] += 1
for (n, c), v in byop.most_common(20):
    print('%5d  %-40s %s' % (v, n, c))
ops = collections.Counter(e.name for e in prof.events() if e.name.startswith('aten::') and any(k in e.name for k in ('copy', 'clone', 'contiguous')))
print(ops.most_common(10))
tops = collections.Counter()
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous'):
        st = [fr for fr in (e.stack or []) if 'normalizing' in fr or 'autograd' in fr][:3]
        par = e.cpu_parent
        tops[(e.name, par.name if par is not None else '-', ' | '.join(s_[-90:] for s_ in st))] += 1
for k, v in tops.most_common(25):
    print(v, k)
