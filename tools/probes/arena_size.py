"""what the two memsets of a train step cover: the gradient bucket and the zero arena (elements, MB), and who asks the arena for how much.
   python tools/probes/arena_size.py c4"""
import collections, importlib, os, sys, traceback
from types import SimpleNamespace as NS
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
name = sys.argv[1] if len(sys.argv) > 1 else 'c4'
pkg = importlib.import_module(bench.PKG)
nftrain = importlib.import_module(bench.PKG + '.train')
nfdata = importlib.import_module(bench.PKG + '.data')
WS = importlib.import_module(bench.PKG + '.workspace')
cfg = bench.CONFIGS[name]
dev = torch.device('cuda:0')
torch.manual_seed(0); np.random.seed(0)
net = getattr(pkg, cfg['cls'])(cfg['dims'], cfg['datatype'], NS(layers=cfg['layers'], mixtures=cfg['mixtures'])).to(dev)
trainer = nftrain.FlowTrainer(net, graph=False)
y = nfdata.sample(cfg['data'], cfg['batch'], 1234)
if cfg['datatype'] == 'image':
    y = y.reshape((cfg['batch'], ) + cfg['dims'])
y = y.to(dev)
for _ in range(3):
    trainer.train_on_batch(y)
who = collections.Counter()
real = WS.ARENA.zeros
def spy(n, device):
    fr = traceback.extract_stack(limit=4)
    who[' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in fr[:-1][-2:])] += (int(n) + 3) & ~3
    return real(n, device)
WS.ARENA.zeros = spy
trainer.train_on_batch(y)
torch.cuda.synchronize()
print('bucket: %d elements = %.1f MB' % (trainer.bucket.flat.numel(), trainer.bucket.flat.numel() * 4 / 1e6))
print('arena: handed out %d elements = %.1f MB per step, zeroed %d = %.1f MB' % (WS.ARENA.last, WS.ARENA.last * 4 / 1e6, WS.ARENA.zeroed, WS.ARENA.zeroed * 4 / 1e6))
for k, v in who.most_common(12):
    print('  %10.2f MB  %s' % (v * 4 / 1e6, k))
