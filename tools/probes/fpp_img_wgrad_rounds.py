"""Flowpp((3, 32, 32), 'image', layers) training step (hipGraph) with the conditioners' weight gradients per coupling and deferred with
several slab rules (fused_flowpp_img.WGRAD_ROUNDS: how many times over a launch of sixteen convolutions fills the chip).
    python tools/probes/fpp_img_wgrad_rounds.py [layers] [batch] [steps]"""
import importlib, os, sys, time
from types import SimpleNamespace as NS
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
pkg = importlib.import_module(bench.PKG)
nftrain = importlib.import_module(bench.PKG + '.train')
nfdata = importlib.import_module(bench.PKG + '.data')
fpi = importlib.import_module(bench.PKG + '.fused_flowpp_img')
dev = torch.device('cuda:0')
y = nfdata.sample('cifar', B, 1234).reshape(B, 3, 32, 32).to(dev)
for label, on, rounds in (('per coupling', False, 0), ('deferred', True, 1), ('deferred', True, 2), ('deferred', True, 4), ('deferred', True, 8)):
    fpi.FPP_IMG_DEFER_ON = on
    fpi.WGRAD_ROUNDS = max(rounds, 1)
    torch.manual_seed(0); np.random.seed(0)
    net = pkg.Flowpp((3, 32, 32), 'image', NS(layers=layers, mixtures=8)).to(dev)
    trainer = nftrain.FlowTrainer(net, graph=True, warmup=2)
    for _ in range(5):
        z, loss = trainer.train_on_batch(y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        z, loss = trainer.train_on_batch(y)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    print('%-13s rounds=%d layers=%d B=%d  %.2f ms/step  %.0f samples/s  loss %.4f' % (label, rounds, layers, B, ms, B / ms * 1e3, float(loss)), flush=True)
    del trainer, net
    torch.cuda.empty_cache()
