"""host-side cost of FlowTrainer.train_on_batch in hipGraph mode (C1: the device step is ~1.07 ms since round 5; is the host slower?)
    python tools/probes/host_step_cost.py [c1|c2|c5]"""
import importlib, sys, time
from types import SimpleNamespace as NS
import numpy as np, torch
sys.path.insert(0, '.')
import bench
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
train = importlib.import_module('normalizing-flows-pytorch_amd.train')
data = importlib.import_module('normalizing-flows-pytorch_amd.data')
N = pkg._native
name = sys.argv[1] if len(sys.argv) > 1 else 'c1'
cfg = bench.CONFIGS[name]
torch.manual_seed(0); np.random.seed(0)
net = getattr(pkg, cfg['cls'])(cfg['dims'], cfg['datatype'], NS(layers=cfg['layers'], mixtures=cfg['mixtures'])).to('cuda')
tr = train.FlowTrainer(net, graph=True, warmup=2)
y = data.sample(cfg['data'], cfg['batch'], 1234).to('cuda')
for _ in range(6):
    tr.train_on_batch(y)
torch.cuda.synchronize()
assert tr._g_fb is not None

def timeit(fn, n=300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6

print('%-34s host enqueue %8.1f us   incl. device %8.1f us' % (('train_on_batch', ) + timeit(lambda: tr.train_on_batch(y))))
print('%-34s host enqueue %8.1f us   incl. device %8.1f us' % (('graph replay only', ) + timeit(lambda: tr._g_fb.replay())))
print('%-34s host enqueue %8.1f us   incl. device %8.1f us' % (('net.train()', ) + timeit(lambda: net.train())))
print('%-34s host enqueue %8.1f us   incl. device %8.1f us' % (('static_y.copy_(y)', ) + timeit(lambda: tr._static_y.copy_(y, non_blocking=True))))
print('%-34s host enqueue %8.1f us   incl. device %8.1f us' % (('check_persistent', ) + timeit(lambda: N.check_persistent())))
