"""phase stamps of the large-batch weight-gradient kernel (k_conv3_bulk_wgrad, workgroup 0 of layer 0: walking wave 0 and filling wave 4,
tiles 2 .. 9 of the workgroup); csrc/conv_bulk.hip built with -DNF_CB_PROF=1 next to conv_bn.hip into build/libcbprof.so:
    python tools/probes/wgrad_prof.py --build ;  python tools/probes/wgrad_prof.py [B] [H] [layers]"""
import ctypes, importlib, os, subprocess, sys
import torch
sys.path.insert(0, '.')
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
N = importlib.import_module('normalizing-flows-pytorch_amd._native')
fc = importlib.import_module('normalizing-flows-pytorch_amd.fused_conv')
here = os.path.dirname(os.path.abspath(pkg.__file__))
lib_path = os.path.join(here, 'build', 'libcbprof.so')
if '--build' in sys.argv:
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=on', '-DNF_CB_PROF=1',
                           '-I' + os.path.join(here, '..', 'include'), '-shared', '-o', lib_path, os.path.join(here, 'csrc', 'conv_bn.hip'),
                           os.path.join(here, 'csrc', 'conv_bulk.hip')])
    print('built', lib_path)
    sys.exit(0)
prof = ctypes.CDLL(lib_path)
real = N.load()
fn = real.nf_conv_bn_wgrad_multi
prof.nf_conv_bn_wgrad_multi.argtypes, prof.nf_conv_bn_wgrad_multi.restype = fn.argtypes, fn.restype
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H = int(sys.argv[2]) if len(sys.argv) > 2 else 16
L = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = 'cuda'
R = fc.R
slabs = int(real.nf_conv_wgrad_slabs(B, H, H, L))
keep = []
arr = (fc.ConvBwdDesc * L)()
scratch = torch.empty(L * slabs * 9216, device=dev)
for i in range(L):
    t = dict(in_=torch.randn(B, 32, H, H, device=dev), gn_src=torch.randn(B, 32, H, H, device=dev), out=torch.randn(B, 32, H, H, device=dev),
             g_skip=torch.randn(B, 32, H, H, device=dev), weight=torch.randn(32, 32, 3, 3, device=dev),
             bn_gamma=torch.rand(32, device=dev) + 0.5, bn_beta=torch.randn(32, device=dev), bn_save_mean=torch.zeros(32, device=dev), bn_save_invstd=torch.ones(32, device=dev),
             cbn_gamma=torch.ones(32, device=dev), cbn_save_mean=torch.zeros(32, device=dev), cbn_save_invstd=torch.ones(32, device=dev),
             cbn_sum_g=torch.zeros(R * 32, device=dev), cbn_sum_gx=torch.zeros(R * 32, device=dev), g_bias=torch.zeros(R * fc.GB, device=dev),
             g_weff=scratch[i * slabs * 9216:(i + 1) * slabs * 9216])
    keep.append(t)
    d = fc._desc(fc.ConvBwdDesc, **t)
    ctypes.memmove(ctypes.addressof(arr) + i * ctypes.sizeof(fc.ConvBwdDesc), ctypes.addressof(d), ctypes.sizeof(fc.ConvBwdDesc))
for _ in range(3):
    rc = prof.nf_conv_bn_wgrad_multi(ctypes.addressof(arr), L, B, 32, 32, H, H, 3, N.stream())
    assert rc == 0, rc
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
prof.nf_cb_prof_read(buf)
t = [v / 100.0 for v in buf]
print('B %d %dx%d, %d layers, %d slabs per layer (us)' % (B, H, H, L, slabs))
for k in range(8):
    w0, w1, w2 = t[3 * k], t[3 * k + 1], t[3 * k + 2]
    f0, f1, f2, f3 = t[24 + 4 * k:28 + 4 * k]
    print('tile %d: walker: walk %.2f | barrier wait %.2f      filler: convert (incl. waiting for its loads) %.2f | issue %.2f | barrier wait %.2f   (tile period %.2f)'
          % (k + 2, w1 - w0, w2 - w1, f1 - f0, f2 - f1, f3 - f2, (t[3 * (k + 1)] - w0) if k < 7 else float('nan')))
