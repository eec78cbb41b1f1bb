"""phase stamps of the one-workgroup RealNVP kernels (csrc/flow_solo.hip built with -DNF_SO_PROF=1 next to the files it links against
into build/libsoprof.so):   python tools/probes/solo_prof.py --build ;  python tools/probes/solo_prof.py [B]"""
import ctypes, importlib, os, subprocess, sys
from types import SimpleNamespace as NS
import torch
sys.path.insert(0, '.')
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
N = pkg._native
here = os.path.dirname(os.path.abspath(pkg.__file__))
lib_path = os.path.join(here, 'build', 'libsoprof.so')
if '--build' in sys.argv:
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=on',
                           '-DNF_SO_NO_WGRAD=1' if '--no-wgrad' in sys.argv else '-DNF_SO_PROF=1',
                           '-I' + os.path.join(here, '..', 'include'), '-shared', '-o', lib_path] +
                          [os.path.join(here, 'csrc', f) for f in ('flow_solo.hip', 'mlp_chain.hip', 'made_chain.hip', 'conv_chain.hip')])
    print('built', lib_path)
    sys.exit(0)
prof = ctypes.CDLL(lib_path)
real = N.load()
for name in ('nf_realnvp_flow_vec_fwd', 'nf_realnvp_flow_vec_bwd_deferred', 'nf_realnvp_flow_pack'):
    fn, pf = getattr(real, name), getattr(prof, name)
    pf.argtypes, pf.restype = fn.argtypes, fn.restype
    setattr(real, name, pf)
train = importlib.import_module('normalizing-flows-pytorch_amd.train')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
net = pkg.RealNVP((2, ), 'density', NS(layers=16, mixtures=8)).to('cuda').train()
tr = train.FlowTrainer(net, graph=False)
y = (torch.randn(B, 2) * 0.7).to('cuda')
for _ in range(3):
    tr._forward_backward(y)
torch.cuda.synchronize()
if not hasattr(prof, 'nf_so_prof_read'):                 # the --no-wgrad build: device time of the two launches (HIP events), no stamps
    for name in ('nf_realnvp_flow_vec_fwd', 'nf_realnvp_flow_vec_bwd_deferred'):
        with N.timed_launches(name) as tl:
            for _ in range(10):
                tr._forward_backward(y)
            d = tl.durations_us()
        print('%-36s %d launches of %d steps: median %.1f us (%.2f us per flow step)' % (name, len(d), 16, sorted(d)[len(d) // 2], sorted(d)[len(d) // 2] / 16))
    sys.exit(0)
buf = (ctypes.c_longlong * 64)()
prof.nf_so_prof_read(buf)
t = [v / 100.0 for v in buf]
print('forward step (us): stage + flow BatchNorm %.2f | head + linear 0 %.2f | BN0 %.2f' % (t[1] - t[0], t[2] - t[1], t[3] - t[2]))
for l in range(1, 5):
    print('   linear %d %.2f | BN%d %.2f' % (l, t[2 + 2 * l] - t[1 + 2 * l], l, t[3 + 2 * l] - t[2 + 2 * l]))
print('   linear 5 + coupling %.2f | end barrier %.2f | step %.2f' % (t[12] - t[11], t[13] - t[12], t[13] - t[0]))
print('backward step (us): stage %.2f | recompute %.2f | linear 5 + coupling %.2f' % (t[33] - t[32], t[34] - t[33], t[35] - t[34]))
prev = t[35]
for J in range(4, -1, -1):
    b = 36 + 4 * (4 - J)
    print('   J = %d: mask + sums %.2f | meeting %.2f | G %.2f | weight gradient %.2f' % (J, t[b] - prev, t[b + 1] - t[b], t[b + 2] - t[b + 1], (t[b + 3] - t[b + 2]) if J >= 1 else 0.0), end='')
    nxt = t[b + 4] if J >= 1 else t[56]
    print(' | data gradient (+ next mask) %.2f' % (nxt - (t[b + 3] if J >= 1 else t[b + 2])))
    prev = nxt if J >= 1 else prev
print('   head %.2f | end barrier %.2f | step %.2f' % (t[57] - t[56], t[58] - t[57], t[58] - t[32]))
