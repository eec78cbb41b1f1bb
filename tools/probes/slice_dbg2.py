"""one slice of a bench config (deterministic weights as tests/test_gpu_slices.py) on several GPU paths vs the float64 oracle
   python tools/probes/slice_dbg2.py c5 16 20"""
import importlib, os, sys
from types import SimpleNamespace as NS
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import trajectory as traj
import bench
pkg = importlib.import_module(bench.PKG)
F = importlib.import_module(bench.PKG + '.fused')
nfdata = importlib.import_module(bench.PKG + '.data')
nftrain = importlib.import_module(bench.PKG + '.train')
cfg = bench.CONFIGS[sys.argv[1]]
a, b = int(sys.argv[2]), int(sys.argv[3])
kind, dims, dt, layers, mix, B = cfg['kind'], cfg['dims'], cfg['datatype'], cfg['layers'], cfg['mixtures'], cfg['batch']
torch.manual_seed(0); np.random.seed(0)
net = getattr(pkg, cfg['cls'])(dims, dt, NS(layers=layers, mixtures=mix))
y = nfdata.sample(cfg['data'], B, 1234)
if cfg['data'] == 'cifar': y = y.reshape((B,) + dims)
sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
_, ora = traj.run(kind, dims, dt, layers, sd0, y, 1, mixtures=mix)
sd0.update({k: v.detach().clone() for k, v in ora.sd.items() if k.endswith(('log_scale', 'bias')) and k.count('.') == 3 and k in sd0})
net.load_state_dict(sd0)
for m in net.modules():
    if hasattr(m, 'initialized'): m.initialized = True
net = net.to('cuda')
trainer = nftrain.FlowTrainer(net, graph=False)
sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
r64 = traj.run_slice(kind, dims, dt, layers, sd, a, b, None, None, mixtures=mix, dtype=torch.float64, y=y)
for mode in ('default', 'steps1', 'layers'):
    net.load_state_dict(sd)
    old = (pkg.Compose.fuse, F.GLOW_FLOW, F.MAF_FLOW)
    if mode == 'layers': pkg.Compose.fuse = False
    if mode == 'steps1': F.GLOW_FLOW, F.MAF_FLOW = '0', False
    z_in = r64['z_in'].float().cuda().requires_grad_(True); ld_in = r64['ld_in'].float().cuda()
    net.train()
    def fl():
        z, ld = net.forward_slice(z_in, ld_in.clone(), a, b)
        return z, nftrain.nll_loss(z, ld)
    z, loss = trainer._run_step(z_in.device, fl)
    pkg.Compose.fuse, F.GLOW_FLOW, F.MAF_FLOW = old
    g = z_in.grad.detach().double().cpu(); w = r64['g_in']
    err = (g - w).abs().reshape(B, -1).max(1).values / w.abs().max()
    bad = {}
    for k, want in r64['grads'].items():
        p = dict(net.named_parameters())[k]
        er = float((p.grad.detach().cpu().double() - want).abs().max() / max(1.0, float(want.abs().max())))
        if er > 1e-4: bad[k] = round(er, 5)
    print(mode, 'z err %.2e' % float((z.detach().cpu().double() - r64['z']).abs().max()), 'g_in rel err max %.2e' % float(err.max()), 'rows bad', int((err > 1e-4).sum()), 'bad params', list(bad.items())[:4])
