"""micro-benchmark of nf_linear_bn_fwd / bwd at the bench shape, timed as hipGraph replays (no host launch cost)."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
F = importlib.import_module('normalizing-flows-pytorch_amd.fused')
N_ = pkg._native
N_.load()
dev = 'cuda'


def graph_us(fn, per_graph=50, replays=10):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per_graph):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(replays):
        g.replay()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / (per_graph * replays)


def main():
    for N in (256, 4096, 16384, 65536):
        x, res, out, gn_src, gn_out, gst = (torch.randn(N, 32, device=dev) for _ in range(6))
        W, g, b = torch.randn(32, 32, device=dev) * 0.2, torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
        gamma, beta = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
        ws = torch.zeros(64, 32, device=dev)
        ws[8] += N
        ws[24] += 1
        rm, rv, nbt = torch.zeros(32, device=dev), torch.ones(32, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
        d = F._desc(F.LinearDesc, in_=x, weight=W, weight_g=g, bias=b, residual=res, out=out, bn_gamma=gamma, bn_beta=beta,
                    bn_sum=ws[0], bn_sqsum=ws[8], bn_center=b, bn_running_mean=rm, bn_running_var=rv, bn_num_batches=nbt,
                    bn_save_mean=ws[16], bn_save_invstd=ws[24], stat_sum=ws[32], stat_sqsum=ws[40])
        fwd = graph_us(lambda: F._launch_fwd([d], N, 32, 32, 1))
        gweff = torch.empty(F.bwd_slabs(N) * 1024, device=dev)
        acc = torch.zeros(64, 32, device=dev)
        db = F._desc(F.LinearBwdDesc, in_=x, weight=W, weight_g=g, bn_gamma=gamma, bn_beta=beta, bn_save_mean=ws[16],
                     bn_save_invstd=ws[24], gn_src=gn_src, out=out, cbn_gamma=gamma, cbn_save_mean=ws[16],
                     cbn_save_invstd=ws[24], cbn_sum_g=acc[0], cbn_sum_gx=acc[8], g_bias=acc[16], g_weff=gweff, gn_out=gn_out,
                     sum_g=acc[24], sum_gx=acc[32])
        bwd = graph_us(lambda: F._launch_bwd([db], N, 32, 32))
        db2 = F._desc(F.LinearBwdDesc, in_=x, weight=W, weight_g=g, bn_gamma=gamma, bn_beta=beta, bn_save_mean=ws[16],
                      bn_save_invstd=ws[24], gn_src=gn_src, out=out, g_skip=res, g_store=gst, cbn_gamma=gamma,
                      cbn_save_mean=ws[16], cbn_save_invstd=ws[24], cbn_sum_g=acc[0], cbn_sum_gx=acc[8], g_bias=acc[16],
                      g_weff=gweff, gn_out=gn_out, sum_g=acc[24], sum_gx=acc[32])
        bwd2 = graph_us(lambda: F._launch_bwd([db2], N, 32, 32))
        one = torch.zeros(1, device=dev)
        tiny = graph_us(lambda: one.add_(1.0))
        print('N=%6d  fwd(all) %6.2f us   bwd %6.2f us   bwd(+skip,+store) %6.2f us   [trivial kernel %5.2f us]' %
              (N, fwd, bwd, bwd2, tiny))


if __name__ == '__main__':
    main()


def variants():
    N = 4096
    x, res, out = (torch.randn(N, 32, device=dev) for _ in range(3))
    W, g, b = torch.randn(32, 32, device=dev) * 0.2, torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
    gamma, beta = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
    ws = torch.zeros(64, 32, device=dev)
    ws[8] += N
    rm, rv, nbt = torch.zeros(32, device=dev), torch.ones(32, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
    parts = {
        'plain': dict(),
        'wn': dict(weight_g=g),
        'bn': dict(bn_gamma=gamma, bn_beta=beta, bn_sum=ws[0], bn_sqsum=ws[8], bn_center=b, bn_running_mean=rm,
                   bn_running_var=rv, bn_num_batches=nbt, bn_save_mean=ws[16], bn_save_invstd=ws[24]),
        'stats': dict(stat_sum=ws[32], stat_sqsum=ws[40]),
        'res': dict(residual=res),
    }
    parts['all'] = {k: v for d in parts.values() for k, v in d.items()}
    for name, kw in parts.items():
        d = F._desc(F.LinearDesc, in_=x, weight=W, bias=b, out=out, **kw)
        print('fwd N=4096 %-6s train %6.2f us   eval %6.2f us' % (name, graph_us(lambda: F._launch_fwd([d], N, 32, 32, 1)),
                                                                 graph_us(lambda: F._launch_fwd([d], N, 32, 32, 0))))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'variants':
    variants()
