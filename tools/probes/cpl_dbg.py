"""fused image coupling (chain launch with the coupling in its epilogue) vs the unfused path: which configs agree / time out"""
import copy, importlib, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
pkg = importlib.import_module(bench.PKG)
fc = importlib.import_module(bench.PKG + '.fused_conv')
N = importlib.import_module(bench.PKG + '._native')
DEV = 'cuda'
for dims, masking, odd, B in [((12, 16, 16), 'channelwise', False, 16), ((12, 16, 16), 'channelwise', False, 64),
                              ((3, 32, 32), 'checkerboard', False, 64), ((48, 8, 8), 'channelwise', False, 64),
                              ((48, 8, 8), 'checkerboard', True, 64)]:
    torch.manual_seed(3)
    k1 = pkg.AffineCoupling(dims, masking=masking, odd=odd).to(DEV)
    k1.net.fused = True
    z = torch.randn((B, ) + dims, device=DEV)
    for mode in ('nograd', 'grad'):
        fc.CONV_COUPLING_ON = True
        N.persistent_reset(1 << 22)
        if mode == 'nograd':
            with torch.no_grad():
                y1, l1 = k1(z, torch.zeros(B, device=DEV))
        else:
            y1, l1 = k1(z.clone().requires_grad_(True), torch.zeros(B, device=DEV))
        torch.cuda.synchronize()
        t1 = N.persistent_timeouts()
        fc.CONV_COUPLING_ON = False
        N.persistent_reset(1 << 22)
        with torch.no_grad():
            y2, l2 = k1(z, torch.zeros(B, device=DEV))
        torch.cuda.synchronize()
        print(dims, masking, odd, B, mode, 'timeouts fused %d unfused %d' % (t1, N.persistent_timeouts()),
              'dy %.2e dl %.2e' % (float((y1 - y2).abs().max()), float((l1 - l2).abs().max())), flush=True)
