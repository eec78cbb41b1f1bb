"""Flowpp((3, 32, 32), 'image') training step on one GPU: ms / step with the HIP conditioner (csrc/flowpp_img.hip) and with the
module stack (NF_FLOWPP_IMG=0), eager and graph-replayed.   python tools/probes/flowpp_img_step.py [layers] [batch] [steps]
For a kernel trace:  rocprofv3 --kernel-trace --stats -d out -- python tools/probes/flowpp_img_step.py 2 64 6 fused"""
import importlib, os, sys, time
from types import SimpleNamespace as NS
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
only = sys.argv[4] if len(sys.argv) > 4 else None
pkg = importlib.import_module(bench.PKG)
nftrain = importlib.import_module(bench.PKG + '.train')
nfdata = importlib.import_module(bench.PKG + '.data')
fpi = importlib.import_module(bench.PKG + '.fused_flowpp_img')
dev = torch.device('cuda:0')
y = nfdata.sample('cifar', B, 1234).reshape(B, 3, 32, 32).to(dev)
for label, on in (('fused', True), ('modules', False)):
    if only and only != label:
        continue
    fpi.FLOWPP_IMG_ON = on
    for graph in (False, True):
        torch.manual_seed(0); np.random.seed(0)
        net = pkg.Flowpp((3, 32, 32), 'image', NS(layers=layers, mixtures=8)).to(dev)
        trainer = nftrain.FlowTrainer(net, graph=graph, warmup=2)
        for _ in range(4):
            z, loss = trainer.train_on_batch(y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            z, loss = trainer.train_on_batch(y)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        print('%-8s graph=%d layers=%d B=%d  %.2f ms/step  %.0f samples/s  loss %.4f' % (label, graph, layers, B, ms, B / ms * 1e3, float(loss)))
