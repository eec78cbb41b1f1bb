"""micro-benchmark of nf_linear_bn_fwd / bwd variants (HIP events on the launch stream)."""
import importlib, sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
F = importlib.import_module('normalizing-flows-pytorch_amd.fused')
N_ = pkg._native
N_.load()
dev = 'cuda'


def timeit(fn, reps=200):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        fn()
    st = torch.cuda.current_stream()
    s.record(st)
    for _ in range(reps):
        fn()
    e.record(st)
    e.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for N in (4096, 65536, 1048576):
    x = torch.randn(N, 32, device=dev)
    out = torch.empty(N, 32, device=dev)
    res = torch.randn(N, 32, device=dev)
    W = torch.randn(32, 32, device=dev) * 0.2
    g = torch.rand(32, device=dev) + 0.5
    b = torch.randn(32, device=dev) * 0.1
    gamma, beta = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
    ws = torch.zeros(8, 32, device=dev)
    ws[1] += 1.0 * N
    rm, rv = torch.zeros(32, device=dev), torch.ones(32, device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    variants = {
        'plain': dict(),
        'wn': dict(weight_g=g),
        'bn': dict(bn_gamma=gamma, bn_beta=beta, bn_sum=ws[0], bn_sqsum=ws[1], bn_center=b, bn_running_mean=rm,
                   bn_running_var=rv, bn_num_batches=nbt, bn_save_mean=ws[2], bn_save_invstd=ws[3]),
        'stats': dict(stat_sum=ws[4], stat_sqsum=ws[5]),
        'res': dict(residual=res),
    }
    variants['all'] = {k: v for d in variants.values() for k, v in d.items()}
    for name, kw in variants.items():
        d = F._desc(F.LinearDesc, in_=x, weight=W, bias=b, out=out, **kw)
        for tr in (1, 0):
            us = timeit(lambda: F._launch_fwd([d], N, 32, 32, tr))
            gbs = N * 32 * 4 * (3 if 'residual' in kw else 2) / us / 1e3
            print('N=%8d fwd %-6s training=%d  %8.2f us  %7.1f GB/s' % (N, name, tr, us, gbs))
    # python-side launch cost (no GPU work): descriptor build + ctypes call
    import time
    t0 = time.perf_counter()
    for _ in range(1000):
        d = F._desc(F.LinearDesc, in_=x, weight=W, bias=b, out=out, **variants['all'])
    print('desc build us', (time.perf_counter() - t0) * 1e3)
