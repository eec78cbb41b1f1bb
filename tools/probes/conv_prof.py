"""phase stamps of workgroup 0 of the fused convolution kernels (csrc/conv_bn.hip built with -DNF_CV_PROF=1 into build/)."""
import ctypes, importlib, os, subprocess, sys
import torch
sys.path.insert(0, '.')
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
N = importlib.import_module('normalizing-flows-pytorch_amd._native')
cond = importlib.import_module('normalizing-flows-pytorch_amd.conditioners')
here = os.path.dirname(os.path.abspath(pkg.__file__))
lib_path = os.path.join(here, 'build', 'libcvprof.so')
if '--build' in sys.argv:
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=on',
                           '-DNF_CV_PROF=1', '-shared', '-o', lib_path, os.path.join(here, 'csrc', 'conv_bn.hip')])
    print('built', lib_path)
    sys.exit(0)
prof = ctypes.CDLL(lib_path)
real = N.load()
# route the conv entry points of the package through the profiled library
for name in ('nf_conv_bn_fwd', 'nf_conv_bn_bwd', 'nf_slab_sum', 'nf_conv_bn_usable', 'nf_conv_bwd_slabs'):
    fn = getattr(real, name)
    pf = getattr(prof, name)
    pf.argtypes, pf.restype = fn.argtypes, fn.restype
    setattr(real, name, pf)
I, O, H, W = [int(v) for v in sys.argv[1:5]]
net = cond.ConvNet(I, O).cuda()
net.fused = True
x = torch.randn(64, I, H, W, device='cuda', requires_grad=True)
for _ in range(3):
    y = net(x); y.sum().backward()
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 32)()
prof.nf_cv_prof_read(buf)
t = [v / 100.0 for v in buf]   # us
names_f = ['consts', 'staging', 'K loop', 'exchange+epilogue', 'stats']
print('fwd (last launch = 1x1 out conv):', ' | '.join('%s %.1f' % (n, t[i + 1] - t[i]) for i, n in enumerate(names_f)))
fc = importlib.import_module('normalizing-flows-pytorch_amd.fused_conv')
x32 = torch.randn(64, 32, H, W, device='cuda'); w = torch.randn(32, 32, 3, 3, device='cuda'); b = torch.randn(32, device='cuda')
o = torch.empty_like(x32); st = torch.zeros(2, 8, 32, device='cuda')
for _ in range(2):
    fc._fwd((64, H, W), 32, 32, 3, True, in_=x32, weight=w, bias=b, out=o, stat_sum=st[0], stat_sqsum=st[1])
torch.cuda.synchronize()
prof.nf_cv_prof_read(buf)
t = [v / 100.0 for v in buf]
print('fwd 3x3 32->32:', ' | '.join('%s %.1f' % (n, t[i + 1] - t[i]) for i, n in enumerate(names_f)))
print('   staging detail: decode %.1f | issue loads %.1f | sync %.1f | w store %.1f | act store %.1f | sync %.1f' % (t[20] - t[1], t[21] - t[20], t[22] - t[21], t[23] - t[22], t[24] - t[23], t[2] - t[24]))
names_b = ['consts', 'staging', 'weight grad', 'data grad K', 'slab+exchange+epilogue', 'sums']
tb = t[8:15]
print('bwd (last launch = first conv):', ' | '.join('%s %.1f' % (n, tb[i + 1] - tb[i]) for i, n in enumerate(names_b)))
