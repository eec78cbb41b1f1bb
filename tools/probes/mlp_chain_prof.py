"""phase timing of the persistent MLP kernels: builds csrc/mlp_chain.hip with -DNF_MC_PROF=1 into build/libmcprof.so
(workgroup 0 stamps wall_clock64, 100 MHz, at phase boundaries) and prints the deltas.
   python tools/probes/mlp_chain_prof.py build      # here (hipcc cross-compiles)
   python tools/probes/mlp_chain_prof.py [N]        # on the GPU box"""
import ctypes
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, 'normalizing-flows-pytorch_amd')
SO = os.path.join(PKG, 'build', 'libmcprof.so')

if len(sys.argv) > 1 and sys.argv[1] == 'build':
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-DNF_MC_PROF=1', '-o', SO,
                           os.path.join(PKG, 'csrc', 'mlp_chain.hip'), os.path.join(PKG, 'csrc', 'made_chain.hip'),
                           os.path.join(PKG, 'csrc', 'conv_chain.hip')])
    print('built', SO)
    sys.exit(0)

import torch

sys.path.insert(0, ROOT)
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
fused = importlib.import_module(pkg.__name__ + '.fused')
cond = importlib.import_module(pkg.__name__ + '.conditioners')
N_ = importlib.import_module(pkg.__name__ + '._native')
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = ctypes.CDLL(SO)
torch.manual_seed(0)
mlp = cond.MLP(1, 2).to(dev).train()
ts = fused._mlp_tensors(mlp)
x = torch.randn(n, 1, device=dev)
out = torch.empty(n, 2, device=dev)
gout = torch.randn(n, 2, device=dev)
gx = torch.empty_like(x)
save = torch.empty(5, 2, 32, device=dev)
tab = fused._ptr_table([t.detach() for t in ts])
learn = list(ts[:18]) + [t for j in range(5) for t in ts[18 + 5 * j:18 + 5 * j + 2]]
dst = [torch.empty_like(t) for t in learn]
gtab = fused._ptr_table(dst)
slabs = torch.empty(N_.header_constant('NF_MLP_BWD_SLAB_FLOATS'), device=dev)
P = ctypes.c_void_p
F = ctypes.c_float
for it in range(3):
    ws = torch.zeros(N_.header_constant('NF_MLP_WS_FLOATS'), device=dev)
    ws2 = torch.zeros(N_.header_constant('NF_MLP_WS_FLOATS'), device=dev)
    torch.cuda.synchronize()
    rc = lib.nf_mlp_chain_fwd(P(x.data_ptr()), tab, P(out.data_ptr()), P(save.data_ptr()), P(ws.data_ptr()), ctypes.c_int64(n), 1, 2,
                              1, F(1e-5), F(0.1), F(1e-5), P(torch.cuda.current_stream().cuda_stream))
    rc2 = lib.nf_mlp_chain_bwd(P(x.data_ptr()), tab, P(save.data_ptr()), P(gout.data_ptr()), P(gx.data_ptr()), gtab, 0,
                               P(ws2.data_ptr()), P(slabs.data_ptr()), ctypes.c_int64(n), 1, 2, 1, F(1e-5), F(1e-5),
                               P(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert rc == 0 and rc2 == 0, (rc, rc2)
buf = (ctypes.c_longlong * 128)()
assert lib.nf_mlp_chain_prof_read(buf) == 0
t = list(buf)


def us(a, b):
    return (t[b] - t[a]) / 100.0


print('N = %d   (workgroup 0, microseconds)' % n)
print('forward : stage %.2f | x + linear0 %.2f | layers %s | tail %.2f | total %.2f' % (
    us(0, 1), us(1, 2), ' '.join('%.2f' % us(2 + i, 3 + i) for i in range(5)), us(7, 8), us(0, 8)))
print('          layer 1 detail: to exchange entry %.2f | exchange %.2f | rest %.2f' % (us(3, 40), us(40, 41), us(41, 4)))
print('backward: stage %.2f | recompute %.2f | layers 5..0 %s | final barrier %.2f | fold %.2f | total %.2f' % (
    us(64, 65), us(65, 66), ' '.join('%.2f' % us(66 + i, 67 + i) for i in range(6)), us(72, 73), us(73, 74), us(64, 74)))
print('          layer 4 detail: tiles %.2f | sync %.2f | wgrad job %.2f | dgrad + sums %.2f | exchange %.2f | BN bwd %.2f' % (
    us(67, 80), us(80, 81), us(81, 82), us(82, 83), us(83, 84), us(84, 68)))

# ---- the fused vector Glow step (same kernels, GLOW variant) ------------------------------------------------------------
a, c, k = pkg.ActNorm((2, )), pkg.InvertibleConv1x1(2), pkg.AffineCoupling((2, ))
mods = torch.nn.ModuleList([a, c, k]).to(dev).train()
head = [a.log_scale, a.bias, c.P, c.L, c.U, c.L_mask, c.U_mask, c.sign_s, c.log_s, k.s_log_scale, k.s_bias]
mts = fused._mlp_tensors(k.net)
htab, mtab = fused._ptr_table([t.detach() for t in head]), fused._ptr_table([t.detach() for t in mts])
lh = [head[0], head[1], head[3], head[4], head[8], head[9], head[10]]
lm = list(mts[:18]) + [t for j in range(5) for t in mts[18 + 5 * j:18 + 5 * j + 2]]
dh, dm = [torch.empty_like(t) for t in lh], [torch.empty_like(t) for t in lm]
hg, mg = fused._ptr_table(dh), fused._ptr_table(dm)
z = torch.randn(n, 2, device=dev)
y = torch.empty_like(z)
ld = torch.zeros(n, device=dev)
gy = torch.randn(n, 2, device=dev)
gz = torch.empty_like(z)
st = P(torch.cuda.current_stream().cuda_stream)
for it in range(3):
    ws = torch.zeros(N_.header_constant('NF_MLP_WS_FLOATS'), device=dev)
    ws2 = torch.zeros(N_.header_constant('NF_MLP_WS_FLOATS'), device=dev)
    torch.cuda.synchronize()
    rc = lib.nf_glow_step_vec_fwd(P(z.data_ptr()), P(y.data_ptr()), P(ld.data_ptr()), htab, mtab, P(save.data_ptr()), P(ws.data_ptr()),
                                  ctypes.c_int64(n), 2, 0, 1, F(1e-5), F(0.1), F(1e-5), st)
    rc2 = lib.nf_glow_step_vec_bwd(P(z.data_ptr()), P(gy.data_ptr()), None, P(gz.data_ptr()), htab, mtab, P(save.data_ptr()), hg, mg, 0,
                                   P(ws2.data_ptr()), P(slabs.data_ptr()), ctypes.c_int64(n), 2, 0, 1, F(1e-5), F(1e-5), st)
    torch.cuda.synchronize()
    assert rc == 0 and rc2 == 0, (rc, rc2)
assert lib.nf_mlp_chain_prof_read(buf) == 0
t = list(buf)
print('--- fused Glow step ---')
print('forward : stage %.2f | x + linear0 %.2f | layers %s | tail %.2f | total %.2f' % (
    us(0, 1), us(1, 2), ' '.join('%.2f' % us(2 + i, 3 + i) for i in range(5)), us(7, 8), us(0, 8)))
print('backward: stage %.2f | recompute %.2f | layers 5..0 %s | final barrier %.2f | fold %.2f | total %.2f' % (
    us(64, 65), us(65, 66), ' '.join('%.2f' % us(66 + i, 67 + i) for i in range(6)), us(72, 73), us(73, 74), us(64, 74)))

# ---- a whole flow of S such steps in one launch per direction (k_glow_flow_*): the stamps are those of the LAST step run ----
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps_m = []
for i in range(S):
    a, c, k = pkg.ActNorm((2, )), pkg.InvertibleConv1x1(2), pkg.AffineCoupling((2, ), odd=bool(i & 1))
    steps_m.append(torch.nn.ModuleList([a, c, k]).to(dev).train())
recs, sinks, keep = [], [], []
nbytes = lib.nf_glow_flow_step_bytes()
host = (ctypes.c_ubyte * (nbytes * S))()
for i, (a, c, k) in enumerate(steps_m):
    head = [a.log_scale, a.bias, c.P, c.L, c.U, c.L_mask, c.U_mask, c.sign_s, c.log_s, k.s_log_scale, k.s_bias]
    mts = fused._mlp_tensors(k.net)
    lh = [head[0], head[1], head[3], head[4], head[8], head[9], head[10]]
    lm = list(mts[:18]) + [t for j in range(5) for t in mts[18 + 5 * j:18 + 5 * j + 2]]
    dh, dm = [torch.zeros_like(t) for t in lh], [torch.zeros_like(t) for t in lm]
    keep += [dh, dm]
    htab, mtab = fused._ptr_table([t.detach() for t in head]), fused._ptr_table([t.detach() for t in mts])
    hg, mg = fused._ptr_table(dh), fused._ptr_table(dm)
    assert lib.nf_glow_flow_pack(P(ctypes.addressof(host) + i * nbytes), htab, mtab, hg, mg, 2, int(i & 1)) == 0
table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)
ys = torch.empty(S, n, 2, device=dev)
gzs = torch.empty(S, n, 2, device=dev)
saves = torch.empty(S, 320, device=dev)
slabs2 = torch.empty(2 * N_.header_constant('NF_MLP_BWD_SLAB_FLOATS'), device=dev)
for it in range(3):
    ws = torch.zeros(S * N_.header_constant('NF_MLP_WS_FLOATS'), device=dev)
    ws2 = torch.zeros(S * N_.header_constant('NF_MLP_WS_FLOATS'), device=dev)
    ld.zero_()
    torch.cuda.synchronize()
    rc = lib.nf_glow_flow_vec_fwd(P(table.data_ptr()), S, P(z.data_ptr()), P(ys.data_ptr()), P(ld.data_ptr()), P(saves.data_ptr()),
                                  P(ws.data_ptr()), ctypes.c_int64(n), 2, 1, F(1e-5), F(0.1), F(1e-5), st)
    rc2 = lib.nf_glow_flow_vec_bwd(P(table.data_ptr()), S, P(z.data_ptr()), P(ys.data_ptr()), P(gy.data_ptr()), None, P(gzs.data_ptr()),
                                   P(saves.data_ptr()), 1, P(ws2.data_ptr()), P(slabs2.data_ptr()), ctypes.c_int64(n), 2, 1, F(1e-5),
                                   F(1e-5), st)
    torch.cuda.synchronize()
    assert rc == 0 and rc2 == 0, (rc, rc2)
assert lib.nf_mlp_chain_prof_read(buf) == 0
t = list(buf)
print('--- whole flow, %d steps in one launch (last step run) ---' % S)
print('forward : loop top -> body %.2f | stage %.2f | x + linear0 %.2f | layers %s | tail %.2f | body %.2f | sync %.2f' % (
    us(100, 0), us(0, 1), us(1, 2), ' '.join('%.2f' % us(2 + i, 3 + i) for i in range(5)), us(7, 8), us(0, 8), us(101, 102)))
print('backward: loop top -> body %.2f | stage %.2f | recompute %.2f | layers 5..0 %s | final barrier %.2f | fold %.2f | body %.2f | sync %.2f' % (
    us(104, 64), us(64, 65), us(65, 66), ' '.join('%.2f' % us(66 + i, 67 + i) for i in range(6)), us(72, 73), us(73, 74), us(64, 74),
    us(105, 106)))
print('          layer 1 detail: to exchange entry %.2f | exchange %.2f | rest %.2f' % (us(3, 40), us(40, 41), us(41, 4)))

# ---- control: the single-step kernel once more, AFTER the whole-flow section (same process state) ----
for it in range(3):
    ws = torch.zeros(N_.header_constant('NF_MLP_WS_FLOATS'), device=dev)
    torch.cuda.synchronize()
    rc = lib.nf_glow_step_vec_fwd(P(z.data_ptr()), P(y.data_ptr()), P(ld.data_ptr()), htab, mtab, P(save.data_ptr()), P(ws.data_ptr()),
                                  ctypes.c_int64(n), 2, 0, 1, F(1e-5), F(0.1), F(1e-5), st)
    torch.cuda.synchronize()
    assert rc == 0
assert lib.nf_mlp_chain_prof_read(buf) == 0
t = list(buf)
print('--- control: single step again ---')
print('forward : stage %.2f | x + linear0 %.2f | layers %s | tail %.2f | total %.2f' % (
    us(0, 1), us(1, 2), ' '.join('%.2f' % us(2 + i, 3 + i) for i in range(5)), us(7, 8), us(0, 8)))

