"""which parameters of a bench config get their gradient from framework autograd (AccumulateGrad) instead of a direct sink"""
import importlib, os, sys, collections
from types import SimpleNamespace as NS
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module(bench.PKG)
nftrain = importlib.import_module(bench.PKG + '.train')
nfdata = importlib.import_module(bench.PKG + '.data')
name = sys.argv[1] if len(sys.argv) > 1 else 'c4'
cfg = bench.CONFIGS[name]
dev = torch.device('cuda:0')
torch.manual_seed(0); np.random.seed(0)
net = getattr(pkg, cfg['cls'])(cfg['dims'], cfg['datatype'], NS(layers=cfg['layers'], mixtures=cfg['mixtures'])).to(dev)
trainer = nftrain.FlowTrainer(net, graph=False, warmup=2)
y = nfdata.sample(cfg['data'], cfg['batch'], 1234)
y = y.reshape((cfg['batch'], ) + cfg['dims']).to(dev) if cfg['data'] == 'cifar' else y.to(dev)
for _ in range(2):
    trainer.train_on_batch(y)
names = {id(p): k for k, p in net.named_parameters()}
kinds = collections.Counter()
for i in trainer._indirect or []:
    k = names[id(trainer.bucket.params[i])]
    kinds['.'.join(k.split('.')[3:])] += 1
print(len(trainer._indirect or []), 'indirect of', len(trainer.bucket.params))
for k, v in kinds.most_common(30):
    print('%5d  %s' % (v, k))
