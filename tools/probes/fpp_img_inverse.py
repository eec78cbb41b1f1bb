"""sampling (inverse) passes of the image Flow++ bench object, for a kernel census under rocprofv3:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -o st -- python tools/probes/fpp_img_inverse.py [passes]
The window between the two marker launches holds `passes` eval-mode net.backward(z) calls at the bench's batch."""
import importlib, os, sys
from types import SimpleNamespace as NS
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG)
nfdata = importlib.import_module(bench.PKG + '.data')
P = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cfg = bench.CONFIGS['fpp_img']
dev = torch.device('cuda:0')
torch.manual_seed(0); np.random.seed(0)
net = getattr(pkg, cfg['cls'])(cfg['dims'], cfg['datatype'], NS(layers=cfg['layers'], mixtures=cfg['mixtures'])).to(dev)
y = nfdata.sample(cfg['data'], cfg['batch'], 1234).reshape((cfg['batch'], ) + cfg['dims']).to(dev)
net.train()
net(y)                                       # data-dependent initialisation
net.eval()
with torch.no_grad():
    z, _ = net(y)
    x, _ = net.backward(z)
    torch.cuda.synchronize()
    print('round trip max |x - y| = %.3e' % float((x - y).abs().max()))
    torch.cuda._sleep(1000)
    for _ in range(P):
        net.backward(z)
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(P):
        net.backward(z)
    b.record(); b.synchronize()
    print('inverse pass: %.3f ms at B = %d' % (a.elapsed_time(b) / P, cfg['batch']))
    a.record()
    for _ in range(P):
        net(y)
    b.record(); b.synchronize()
    print('forward pass: %.3f ms' % (a.elapsed_time(b) / P))
