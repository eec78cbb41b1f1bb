"""phase stamps of wave 0 of workgroup 0 of the large-batch forward kernel (csrc/conv_bulk.hip built with -DNF_CB_PROF=1 next to
conv_bn.hip into build/libcbprof.so):   python tools/probes/bulk_prof.py --build ;  python tools/probes/bulk_prof.py [B] [H] [nblk]"""
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
                           '-shared', '-o', lib_path, os.path.join(here, 'csrc', 'conv_bn.hip'), os.path.join(here, 'csrc', 'conv_bulk.hip')])
    print('built', lib_path)
    sys.exit(0)
prof = ctypes.CDLL(lib_path)
real = N.load()
for name in ('nf_conv_bn_fwd', 'nf_conv_bn_bwd', 'nf_conv_bulk_config'):
    fn, pf = getattr(real, name), getattr(prof, name)
    pf.argtypes, pf.restype = fn.argtypes, fn.restype
    setattr(real, name, pf)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H = int(sys.argv[2]) if len(sys.argv) > 2 else 16
nblk = int(sys.argv[3]) if len(sys.argv) > 3 else 2
N.call('nf_conv_bulk_config', 1, 0, nblk)
DEV = 'cuda'
x = torch.randn(B, 32, H, H, device=DEV); w = torch.randn(32, 32, 3, 3, device=DEV) * 0.08; b = torch.randn(32, device=DEV)
res = torch.randn(B, 32, H, H, device=DEV)
o = torch.empty_like(x); st = torch.zeros(2, 8 * 32, device=DEV)
g, be, ce = torch.ones(32, device=DEV), torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
s1, s2 = torch.zeros(256, device=DEV), torch.zeros(256, device=DEV); s2[:32] = B * H * H
rm, rv, sm, si = (torch.zeros(32, device=DEV) for _ in range(4))
nimg = int(real.nf_conv_weight_pack_images(32, 32, 3))
pack = torch.empty(nimg * N.header_constant('NF_CONV_PACK_IMAGE_FLOATS'), device=DEV)
d = fc.ConvPackDesc(w.data_ptr(), pack.data_ptr(), 32, 32, 3, 0)
N.call('nf_conv_weight_pack', ctypes.addressof(d), 1, N.stream())
for _ in range(3):
    fc._fwd((B, H, H), 32, 32, 3, True, in_=x, weight=w, bias=b, residual=res, out=o, stat_sum=st[0], stat_sqsum=st[1], wpk=pack, bn_gamma=g, bn_beta=be,
            bn_sum=s1, bn_sqsum=s2, bn_center=ce, bn_running_mean=rm, bn_running_var=rv, bn_save_mean=sm, bn_save_invstd=si)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
prof.nf_cb_prof_read(buf)
t = [v / 100.0 for v in buf]
print('B %d %dx%d nblk %d (us, wave 0 of workgroup 0)' % (B, H, H, nblk))
print('prologue: consts + weights %.2f | items etc + barrier %.2f | first issue %.2f' % (t[1] - t[0], t[2] - t[1], t[3] - t[2]))
pu = 0
while pu < 7 and t[8 + 8 * pu] > 0 and t[4 + 8 * pu] >= t[0]:
    b0 = 4 + 8 * pu
    prev = t[3] if pu == 0 else t[b0 - 4]
    print('unit %d: wait for loads %.2f | convert + frame %.2f | issue next + residual %.2f | K loop %.2f | epilogue %.2f'
          % (pu, t[b0] - prev, t[b0 + 1] - t[b0], t[b0 + 2] - t[b0 + 1], t[b0 + 3] - t[b0 + 2], t[b0 + 4] - t[b0 + 3]))
    pu += 1
print('tail (statistics) %.2f | total %.2f' % (t[61] - t[60], t[61] - t[0]))
