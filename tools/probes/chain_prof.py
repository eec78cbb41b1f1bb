"""phase stamps of workgroup 0 of the persistent ConvNet kernel (csrc/conv_chain.hip built with -DNF_CC_PROF=1 into build/).
   python tools/probes/chain_prof.py --build ;  python tools/probes/chain_prof.py I O H W [B]"""
import ctypes, importlib, os, subprocess, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
N = importlib.import_module('normalizing-flows-pytorch_amd._native')
cond = importlib.import_module('normalizing-flows-pytorch_amd.conditioners')
here = os.path.dirname(os.path.abspath(pkg.__file__))
lib_path = os.path.join(here, 'build', 'libccprof.so')
if '--build' in sys.argv:
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=on',
                           '-DNF_CC_PROF=1', '-shared', '-o', lib_path, os.path.join(here, 'csrc', 'conv_chain.hip')])
    print('built', lib_path)
    sys.exit(0)
prof = ctypes.CDLL(lib_path)
real = N.load()
for name in ('nf_convnet_chain_fwd', 'nf_convnet_chain_bwd', 'nf_convnet_chain_usable', 'nf_convnet_chain_ws_floats', 'nf_conv_weight_pack', 'nf_conv_weight_pack_images'):
    fn = getattr(real, name)
    pf = getattr(prof, name)
    pf.argtypes, pf.restype = fn.argtypes, fn.restype
    setattr(real, name, pf)
I, O, H, W = [int(v) for v in sys.argv[1:5]]
B = int(sys.argv[5]) if len(sys.argv) > 5 else 64
net = cond.ConvNet(I, O).cuda().train()
net.fused = True
fc = importlib.import_module('normalizing-flows-pytorch_amd.fused_conv')
if '--nopack' not in sys.argv:                       # the path a model takes: weight images written once, streamed by the kernel
    wns = [m for m in net.modules() if isinstance(m, cond.WeightNorm)]
    with torch.no_grad():
        for m in wns:
            m._w_eff = m.effective_weight().contiguous()
    fc.pack_conv_weights(wns, [m._w_eff for m in wns])
    assert fc._convnet_packs(net) is not None
x = torch.randn(B, I, H, W, device='cuda')
CPL = '--cpl' in sys.argv                           # with the fused coupling: the launch a model makes (checkerboard split of (I/2, 2H, 2W))
if CPL:
    NF = importlib.import_module('normalizing-flows-pytorch_amd.functional')
    z = torch.randn(B, I // 2, 2 * H, 2 * W, device='cuda')
    a, c = torch.full((1, ), 0.5, device='cuda'), torch.zeros(1, device='cuda')
    BWD = '--bwd' in sys.argv
    HEAD = '--head' in sys.argv                      # the next step's head transposed in the backward launch's prologue (nf_cc_head_bwd)
    Cf = I // 2
    hW, hls = torch.randn(Cf, Cf, device='cuda') / Cf ** 0.5, torch.randn(Cf, device='cuda') * 0.1

    class FakeHead(torch.autograd.Function):
        @staticmethod
        def forward(ctx, y):
            return y.clone()

        @staticmethod
        def backward(ctx, g_h):
            g_x = torch.empty_like(g_h)
            NF.PENDING_HEAD_BWD[g_x.data_ptr()] = (g_h.contiguous(), hls, hW, g_x, None if Cf >= 9 else (N.SPLIT_CHECKER, 0))
            return g_x
    from types import SimpleNamespace as NS
    holder = NS(meta={}, pending=[], g_ld={})
    h_b, h_logs = torch.randn(Cf, device='cuda') * 0.1, torch.randn(Cf, device='cuda') * 0.1
    FWD_HEAD = HEAD and not BWD and 9 <= Cf <= 64    # the step's head in the forward launch's prologue (nf_cc_head_fwd)
    for _ in range(3):
        with torch.set_grad_enabled(BWD):
            zz = z.clone().requires_grad_(BWD)
            ld = torch.zeros(B, device='cuda')
            if FWD_HEAD:
                zz, xx, ld = NF.glow_head_w(zz, ld, hls, h_b, hW, h_logs, holder, 0, N.SPLIT_CHECKER, 0, defer=True)
            else:
                xx = NF.half_gather(zz.detach(), 1, N.SPLIT_CHECKER, 0)
            y, ld2 = fc.convnet_coupling(net, xx, zz, ld, a, c, N.SPLIT_CHECKER, 0)
            assert not NF.PENDING_HEADS
            if BWD:
                if HEAD:
                    y = FakeHead.apply(y)
                (y.square().sum() * 1e-3 + ld2.sum()).backward()
                assert not NF.PENDING_HEAD_BWD
else:
    with torch.no_grad():
        for _ in range(3):
            y = net(x)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 128)()
prof.nf_cc_prof_read(buf)
t = [v / 100.0 for v in buf]   # us
print('B %d  %d -> %d  %d x %d   total %.1f us: zero+conv0 %.1f' % (B, I, O, H, W, t[51] - t[0], t[1] - t[0]))
for l in range(5):
    o = 8 * l
    print('  layer %d: wait/sync %.1f | k-split %.1f | finish+tile stats %.1f | grid exchange %.1f | consts %.1f | normalise+weights %.1f | K loop %.1f'
          % (l, t[2 + o] - (t[1] if l == 0 else t[8 + o - 8]), t[3 + o] - t[2 + o], t[4 + o] - t[3 + o], t[5 + o] - t[4 + o], t[6 + o] - t[5 + o],
             (t[7 + o] - t[6 + o]) if l < 4 else 0.0, (t[8 + o] - t[7 + o]) if l < 4 else 0.0))
print('  1x1 out conv %.1f' % (t[51] - t[50]))
print('  exchange of layer 1: sync %.1f | combine+publish %.1f | poll %.1f | sync %.1f | merge %.1f' % (
    t[56] - t[12], t[57] - t[56], t[58] - t[57], t[59] - t[58], t[13] - t[59]))
if '--head' in sys.argv and '--bwd' not in sys.argv:
    print('  prologue: requests %.1f | zero %.1f | head %.1f [W, An to LDS %.1f | barrier %.1f | blocks %.1f (loads %.1f, normalise + MFMA %.1f, outputs %.1f) | barrier %.1f] | rest of conv0 %.1f'
          % (t[105] - t[0], t[106] - t[105], t[107] - t[106], t[108] - t[106], t[109] - t[108], t[110] - t[109], t[111] - t[109], t[112] - t[111], t[110] - t[112], t[107] - t[110], t[1] - t[107]))
if '--bwd' in sys.argv:
    print('backward total %.1f us: zero + 1x1^T (coupling backward on the fly) %.1f' % (t[97] - t[64], t[65] - t[64]))
    for l in range(4, -1, -1):
        o = 66 + 6 * (4 - l)
        prev = t[65] if l == 4 else t[o - 1]
        print('  layer %d: loads+sync %.1f | k-split %.1f | gn, sums, exchange %.1f | G -> frame %.1f | weights wait+sync %.1f | K loop %.1f'
              % (l, t[o] - prev, t[o + 1] - t[o], t[o + 2] - t[o + 1], t[o + 3] - t[o + 2], (t[o + 4] - t[o + 3]) if l >= 1 else 0.0,
                 (t[o + 5] - t[o + 4]) if l >= 1 else 0.0))
    print('  conv0^T chunks + stores %.1f' % (t[97] - t[96]))
    print('  prologue: start -> head (zero, requests) %.1f | head %.1f [W to LDS %.1f | barrier %.1f | items %.1f | barrier %.1f] | 1x1^T %.1f'
          % (t[100] - t[64], t[101] - t[100], t[102] - t[100], t[103] - t[102], t[104] - t[103], t[101] - t[104], t[65] - t[101]))
arr = (ctypes.c_longlong * 128)()
prof.nf_cc_arrive_read(arr)
G = (B * H * W + (255 if H * W >= 256 else 127)) // (256 if H * W >= 256 else 128)
a = [arr[i] / 100.0 for i in range(G)]
print('  arrival of the workgroups at the poll of layer 1, us after the first: ' + ' '.join('%.1f' % (v - min(a)) for v in a))
if I > 32:
    print('  chunk 1 of conv0: weights requested, barrier %.1f | weights to LDS %.1f | frame loads + stores %.1f | barrier %.1f | K loop %.1f   (chunk 0 began %.1f us before)'
          % (t[53] - t[52], t[54] - t[53], t[55] - t[54], t[60] - t[55], t[61] - t[60], t[52] - t[0]))
