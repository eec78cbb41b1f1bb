"""phase stamps of workgroup 0 of k_glow_head_w_bwd (csrc/glow_head_mfma.hip built with -DNF_GH_PROF=1 into build/):
   python tools/probes/head_prof.py --build ; python tools/probes/head_prof.py C H [B]"""
import ctypes, importlib, os, subprocess, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
N = importlib.import_module('normalizing-flows-pytorch_amd._native')
here = os.path.dirname(os.path.abspath(pkg.__file__))
lib_path = os.path.join(here, 'build', 'libghprof.so')
if '--build' in sys.argv:
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=on', '-DNF_GH_PROF=1',
                           '-shared', '-o', lib_path, os.path.join(here, 'csrc', 'glow_head_mfma.hip')])
    print('built', lib_path)
    sys.exit(0)
prof = ctypes.CDLL(lib_path)
fn = N.load().nf_glow_head_w_bwd
pf = prof.nf_glow_head_w_bwd
pf.argtypes, pf.restype = fn.argtypes, fn.restype
C, H = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
x, gh = torch.randn(B, C, H, H, device='cuda'), torch.randn(B, C, H, H, device='cuda')
W = torch.linalg.qr(torch.randn(C, C))[0].cuda().contiguous()
ls, bs, gld = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda'), torch.randn(B, device='cuda')
gx, gls, gb, gW = torch.empty_like(x), torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda'), torch.zeros(C, C, device='cuda')
for _ in range(3):
    rc = pf(gh.data_ptr(), gld.data_ptr(), x.data_ptr(), ls.data_ptr(), bs.data_ptr(), W.data_ptr(), gx.data_ptr(), gls.data_ptr(), gb.data_ptr(),
            gW.data_ptr(), B, C, H, H, N.stream())
    assert rc == 0, rc
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 16)()
prof.nf_gh_prof_read(buf)
t = [v / 100.0 for v in buf]
print('C %d  %d x %d  B %d: constants, W fragments, first tile requested %.1f | staged, barriers %.1f | g_W MFMAs %.1f | g_a, g_x stores, sums %.1f | '
      'g_W reduction + atomics %.1f | channel sums + atomics %.1f   total %.1f us'
      % (C, H, H, B, t[9] - t[8], t[10] - t[9], t[11] - t[10], t[12] - t[11], t[13] - t[12], t[14] - t[13], t[14] - t[8]))

# ---- forward (stamps 0 .. 5) ----
ff = N.load().nf_glow_head_w_fwd
pff = prof.nf_glow_head_w_fwd
pff.argtypes, pff.restype = ff.argtypes, ff.restype
lsv = torch.zeros(C, device='cuda')
h, z1c, ld = torch.empty_like(x), torch.empty(B, C // 2, H, H, device='cuda'), torch.zeros(B, device='cuda')
for _ in range(3):
    rc = pff(x.data_ptr(), ls.data_ptr(), bs.data_ptr(), W.data_ptr(), lsv.data_ptr(), h.data_ptr(), z1c.data_ptr(), ld.data_ptr(), 2, 0, B, C, H, H,
             N.stream())
    assert rc == 0, rc
torch.cuda.synchronize()
prof.nf_gh_prof_read(buf)
t = [v / 100.0 for v in buf]
print('   forward: log-det + loads issued %.1f | W staged (memory round trip, LDS, barrier) %.1f | fragments from LDS %.1f | ActNorm + MFMAs %.1f | '
      'stores issued %.1f   total %.1f us' % (t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0]))
