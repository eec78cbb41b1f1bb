"""data gradient of ONE flow step (RealNVP / Glow / MAF, 2-D) on the GPU vs the oracle in float32 and float64.
   python tools/probes/step_grad.py realnvp 256"""
import importlib
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import models as om, transforms as tf, trajectory as traj  # noqa: E402

kind = sys.argv[1]
B = int(sys.argv[2])
cls = {'realnvp': 'RealNVP', 'glow': 'Glow', 'maf': 'MAF'}[kind]
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
nfdata = importlib.import_module('normalizing-flows-pytorch_amd.data')
nftrain = importlib.import_module('normalizing-flows-pytorch_amd.train')
for seed in range(4):
    torch.manual_seed(seed)
    np.random.seed(seed)
    net = getattr(pkg, cls)((2, ), '2d', NS(layers=1, mixtures=None))
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    y = nfdata.sample('moons', B, 1234 + seed) * (1.0 + seed) + 0.3 * seed
    out = {}
    for dt in (torch.float32, torch.float64):
        ora = om.FlowOracle(kind, (2, ), '2d', 1, traj.cast_state(sd0, dt), training=True, actnorm_initialized=True).requires_grad_(True)
        yy = y.detach().clone().to(dt).requires_grad_(True)
        z, ld = ora.forward(yy)
        loss = tf.nll_loss(z, ld)
        loss.backward()
        out[dt] = (yy.grad.detach(), z.detach())
    for m in net.modules():
        if hasattr(m, 'initialized'):
            m.initialized = True
    net = net.to('cuda').train()
    res = {}
    for mode in ('layers', 'fused'):
        pkg.Compose.fuse = mode == 'fused'
        net.load_state_dict(sd0)
        yd = y.detach().clone().to("cuda").requires_grad_(True)
        z, ld = net(yd)
        loss = nftrain.nll_loss(z, ld)
        loss.backward()
        res[mode] = (yd.grad.detach().cpu(), z.detach().cpu())
        for p in net.parameters():
            p.grad = None
    pkg.Compose.fuse = True
    g64 = out[torch.float64][0]
    s = float(g64.abs().max())
    print('seed %d  max|g| %.3e   cpu32 %.2e   gpu layers %.2e   gpu fused %.2e   (z: cpu32 %.1e gpu %.1e %.1e)' % (
        seed, s, float((out[torch.float32][0].double() - g64).abs().max()) / s,
        float((res['layers'][0].double() - g64).abs().max()) / s, float((res['fused'][0].double() - g64).abs().max()) / s,
        float((out[torch.float32][1].double() - out[torch.float64][1]).abs().max()),
        float((res['layers'][1].double() - out[torch.float64][1]).abs().max()), float((res['fused'][1].double() - out[torch.float64][1]).abs().max())))
