"""the deferred weight-gradient launch in isolation: sixteen 3x3 32->32 layers of one pyramid level per nf_conv_bn_wgrad_multi call,
operands as the chain backward leaves them; checked against torch autograd's conv2d weight gradient, timed with events.
    python tools/probes/wgrad_time.py H [B] [layers]"""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
N = importlib.import_module('normalizing-flows-pytorch_amd._native')
fc = importlib.import_module('normalizing-flows-pytorch_amd.fused_conv')
H = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 64; L = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = 'cuda'
torch.manual_seed(0)
R = fc.R
slabs = int(N.load().nf_conv_wgrad_slabs(B, H, H, L))
lay = []
for _ in range(L):
    t = dict(x=torch.randn(B, 32, H, H, device=dev), gn=torch.randn(B, 32, H, H, device=dev), out=torch.randn(B, 32, H, H, device=dev),
             skip=torch.randn(B, 32, H, H, device=dev), w=torch.randn(32, 32, 3, 3, device=dev),
             g1=torch.rand(32, device=dev) + 0.5, b1=torch.randn(32, device=dev), m1=torch.randn(32, device=dev) * 0.1, s1=torch.rand(32, device=dev) + 0.5,
             g2=torch.rand(32, device=dev) + 0.5, m2=torch.randn(32, device=dev) * 0.1, s2=torch.rand(32, device=dev) + 0.5,
             sg=torch.zeros(R * 32, device=dev), sgx=torch.zeros(R * 32, device=dev), gb=torch.zeros(R * fc.GB, device=dev))
    t['sg'][:32] = torch.randn(32, device=dev) * B * H * H * 0.01
    t['sgx'][:32] = torch.randn(32, device=dev) * B * H * H * 0.01
    lay.append(t)
scratch = torch.empty(L * slabs * 32 * 32 * 9, device=dev)
g_w = [torch.empty(32, 32, 3, 3, device=dev) for _ in range(L)]
arr = (fc.ConvBwdDesc * L)()
jobs = []
for i, t in enumerate(lay):
    region = scratch[i * slabs * 9216:(i + 1) * slabs * 9216]
    d = fc._desc(fc.ConvBwdDesc, in_=t['x'], weight=t['w'], bn_gamma=t['g1'], bn_beta=t['b1'], bn_save_mean=t['m1'], bn_save_invstd=t['s1'],
                 g_skip=t['skip'], gn_src=t['gn'], out=t['out'], cbn_gamma=t['g2'], cbn_save_mean=t['m2'], cbn_save_invstd=t['s2'],
                 cbn_sum_g=t['sg'], cbn_sum_gx=t['sgx'], g_bias=t['gb'], g_weff=region)
    ctypes.memmove(ctypes.addressof(arr) + i * ctypes.sizeof(fc.ConvBwdDesc), ctypes.addressof(d), ctypes.sizeof(fc.ConvBwdDesc))
    jobs.append((region, g_w[i], 9216, 9216, slabs, False, 9))


def launch():
    N.call('nf_conv_bn_wgrad_multi', ctypes.addressof(arr), L, B, 32, 32, H, H, 3, N.stream())


launch(); fc._slab_sum_all(jobs)
torch.cuda.synchronize()
# reference: G = skip + BatchNorm backward of gn at out; input = relu(bn(x)); weight gradient of conv2d
worst = 0.0
for i in (0, L - 1):
    t = lay[i]
    n = B * H * H
    xh = (t['out'] - t['m2'].view(1, -1, 1, 1)) * t['s2'].view(1, -1, 1, 1)
    G = t['skip'] + (t['g2'] * t['s2']).view(1, -1, 1, 1) * (t['gn'] - (t['sg'][:32] / n).view(1, -1, 1, 1) - xh * (t['sgx'][:32] / n).view(1, -1, 1, 1))
    a = torch.relu((t['x'] - t['m1'].view(1, -1, 1, 1)) * (t['g1'] * t['s1']).view(1, -1, 1, 1) + t['b1'].view(1, -1, 1, 1))
    ref = torch.nn.grad.conv2d_weight(a.double(), (32, 32, 3, 3), G.double(), padding=1)
    err = ((g_w[i].double() - ref).abs().max() / ref.abs().max()).item()
    gb = t['gb'].view(R, fc.GB)[:, :32].sum(0)
    errb = ((gb.double() - G.double().sum((0, 2, 3))).abs().max() / G.double().sum((0, 2, 3)).abs().max()).item()
    worst = max(worst, err, errb / (i + 1) if i == 0 else 0.0)      # (g_bias accumulates over the repeated launches below: checked once)
print('H %d B %d layers %d slabs %d: max rel err vs fp64 conv2d_weight %.2e' % (H, B, L, slabs, worst))
for _ in range(3):
    launch()
e0, e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    launch()
e1.record()
for _ in range(20):
    fc._slab_sum_all(jobs)
e2.record()
torch.cuda.synchronize()
tw, ts = e0.elapsed_time(e1) / 20 * 1e3, e1.elapsed_time(e2) / 20 * 1e3
fl = 2.0 * L * B * H * H * 32 * 288
print('   weight-gradient launch %.1f us (%.1f TFLOP/s, %.1f %% of the 157 TFLOP/s fp32 MFMA peak) + slab sum %.1f us' % (tw, fl / tw / 1e6, fl / tw / 1e6 / 1.57, ts))
if os.environ.get('NF_WGRAD_PROF') == '1':      # phase stamps of workgroup 0, second tile (conv_bn.hip built with -DNF_CV_PROF=1)
    here = os.path.dirname(os.path.abspath(pkg.__file__))
    prof = ctypes.CDLL(os.path.join(here, 'build', 'libcvprof.so'))
    fn = N.load().nf_conv_bn_wgrad_multi
    prof.nf_conv_bn_wgrad_multi.argtypes, prof.nf_conv_bn_wgrad_multi.restype = fn.argtypes, fn.restype
    for _ in range(2):
        rc = prof.nf_conv_bn_wgrad_multi(ctypes.addressof(arr), L, B, 32, 32, H, H, 3, N.stream())
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 32)()
    prof.nf_cv_prof_read(buf)
    t = [v / 100.0 for v in buf]
    print('   workgroup 0: launch, constants, first tile (fill + walk) %.1f us | second tile, barrier to barrier %.1f us' % (t[15] - t[8], t[14] - t[15]))
