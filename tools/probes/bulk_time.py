"""device time of the large-batch 3x3 kernels (csrc/conv_bulk.hip) against the per-layer kernels (csrc/conv_bn.hip) on one layer:
    python tools/probes/bulk_time.py [B] [H] -> us per launch forward / backward data pass, bulk off / nblk 1 / nblk 2"""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
fc = importlib.import_module('normalizing-flows-pytorch_amd.fused_conv')
N = pkg._native
N.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H = int(sys.argv[2]) if len(sys.argv) > 2 else 16
I = int(sys.argv[3]) if len(sys.argv) > 3 else 32
DEV = 'cuda'
R = 8
x = torch.randn(B, I, H, H, device=DEV)
w = torch.randn(32, I, 3, 3, device=DEV) * 0.08
bias = torch.randn(32, device=DEV)
res = torch.randn(B, 32, H, H, device=DEV)
out = torch.empty(B, 32, H, H, device=DEV)
gamma, beta, center = torch.rand(I, device=DEV) + 0.5, torch.randn(I, device=DEV), torch.zeros(I, device=DEV)
s1, s2 = torch.zeros(R * 32, device=DEV), torch.zeros(R * 32, device=DEV)
s2[:I] = float(B * H * H)
st1, st2 = torch.zeros(R * 32, device=DEV), torch.zeros(R * 32, device=DEV)
rm, rv, sm, si = torch.zeros(I, device=DEV), torch.ones(I, device=DEV), torch.zeros(32, device=DEV), torch.ones(32, device=DEV)
nimg = int(N.load().nf_conv_weight_pack_images(32, I, 3))
pack = torch.empty(nimg * N.header_constant('NF_CONV_PACK_IMAGE_FLOATS'), device=DEV)
d = fc.ConvPackDesc(w.data_ptr(), pack.data_ptr(), 32, I, 3, 0)
N.call('nf_conv_weight_pack', ctypes.addressof(d), 1, N.stream())
gn_src, g_skip = torch.randn(B, 32, H, H, device=DEV), torch.randn(B, 32, H, H, device=DEV)
g_store, gn_out = torch.empty(B, 32, H, H, device=DEV), torch.empty(B, I, H, H, device=DEV)
sg, sgx, cs1, cs2 = (torch.zeros(R * 32, device=DEV) for _ in range(4))
cg = torch.ones(32, device=DEV)


def fwd(pk):
    kw = dict(in_=x, weight=w, bias=bias, residual=res, out=out, stat_sum=st1, stat_sqsum=st2, wpk=pk)
    if I == 32:
        kw.update(bn_gamma=gamma, bn_beta=beta, bn_sum=s1, bn_sqsum=s2, bn_center=center, bn_running_mean=rm, bn_running_var=rv, bn_save_mean=sm[:I], bn_save_invstd=si[:I])
    fc._fwd((B, H, H), I, 32, 3, True, **kw)


def bwd(pk):
    kw = dict(in_=x, weight=w, gn_src=gn_src, out=out, g_skip=g_skip, g_store=g_store, gn_out=gn_out, cbn_gamma=cg, cbn_save_mean=sm, cbn_save_invstd=si,
              cbn_sum_g=cs1, cbn_sum_gx=cs2, wpk=pk)
    if I == 32:
        kw.update(bn_gamma=gamma, bn_beta=beta, bn_save_mean=sm, bn_save_invstd=si, sum_g=sg, sum_gx=sgx)
    fc._bwd((B, H, H), I, 32, 3, **kw)


def t(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / (5 * reps)


flop = 2.0 * B * H * H * 9 * I * 32
print('B %d  %dx%d  I %d: %.3f GFLOP per launch' % (B, H, H, I, flop / 1e9))
for name, on, nblk, pk in (('per-layer (conv_bn.hip)', 0, 0, None), ('bulk nblk=1 packed', 1, 1, pack), ('bulk nblk=2 packed', 1, 2, pack), ('bulk nblk=2 unpacked', 1, 2, None)):
    N.call('nf_conv_bulk_config', on, 0, nblk)
    tf_, tb_ = t(lambda: fwd(pk)), t(lambda: bwd(pk))
    print('%-26s forward %7.1f us (%5.1f TF)   backward data %7.1f us (%5.1f TF)' % (name, tf_, flop / tf_ / 1e6, tb_, flop / tb_ / 1e6))
