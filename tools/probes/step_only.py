"""N graph-replayed train steps of one bench config and nothing else (for rocprofv3 --kernel-trace: tools/probes/step_timeline.py
reads the trace).   python tools/probes/step_only.py c4 [steps]"""
import importlib, os, sys
from types import SimpleNamespace as NS
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
name = sys.argv[1] if len(sys.argv) > 1 else 'c4'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pkg = importlib.import_module(bench.PKG)
nftrain = importlib.import_module(bench.PKG + '.train')
nfdata = importlib.import_module(bench.PKG + '.data')
cfg = bench.CONFIGS[name]
dev = torch.device('cuda:0')
torch.manual_seed(0); np.random.seed(0)
net = getattr(pkg, cfg['cls'])(cfg['dims'], cfg['datatype'], NS(layers=cfg['layers'], mixtures=cfg['mixtures'])).to(dev)
trainer = nftrain.FlowTrainer(net, graph=True, warmup=2)
y = nfdata.sample(cfg['data'], cfg['batch'], 1234)
if cfg['datatype'] == 'image':
    y = y.reshape((cfg['batch'], ) + cfg['dims'])
y = y.to(dev)
for _ in range(3 + steps):
    trainer.train_on_batch(y)
    torch.cuda.synchronize()
print('done')
