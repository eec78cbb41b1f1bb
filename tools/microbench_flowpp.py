"""times the fused Flow++ conditioner kernels (csrc/flowpp_cond.hip) with HIP events over graph-free back-to-back launches:
   python tools/microbench_flowpp.py            # N sweep: per-tile slope and fixed cost of forward and backward"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
fused = importlib.import_module(pkg.__name__ + '.fused')
N_ = importlib.import_module(pkg.__name__ + '._native')
dev = torch.device('cuda:0')


def timeit(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    torch.manual_seed(0)
    layer = pkg.MixLogAttnCoupling((2, ), n_mixtures=8).to(dev)
    ts, F_ = fused._flowpp_tensors(layer.net)
    O = ts[13].shape[0]
    grads = [torch.zeros_like(t) for t in ts]
    d = [g.data_ptr() for g in grads]
    d[7] += 4 * 2 * F_ * 32
    d[8] += 4 * 2 * F_
    print('N        fwd_us   bwd_us')
    for n in (32, 4096, 32768, 65536, 131072, 262144, 1048576):
        x = torch.randn(n, 1, device=dev)
        out = torch.empty(n, O, device=dev)
        g_out = torch.randn(n, O, device=dev)
        g_x = torch.empty_like(x)
        args = fused._flowpp_fwd_args(ts, F_)

        def fwd():
            N_.call('nf_flowpp_cond_fwd', N_.ptr(x), *args, N_.ptr(out), 1, 1, n, 1, O, N_.stream())

        def bwd():
            N_.call('nf_flowpp_cond_bwd', N_.ptr(x), *args, N_.ptr(g_out), N_.ptr(g_x), *d,
                    N_.ptr(fused.flowpp_bwd_workspace(dev)), 1, 1, 1, 1, 0, n, 1, O, N_.stream())

        print('%-8d %7.1f  %7.1f' % (n, timeit(fwd), timeit(bwd)))


if __name__ == '__main__':
    main()
