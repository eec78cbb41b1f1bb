"""MLP conditioner forward (and, once wired, backward): multi-launch nf_linear_bn_* chain vs the persistent
nf_mlp_chain_* kernels, timed as hipGraph replays (no host launch cost)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('normalizing-flows-pytorch_amd')
fused = importlib.import_module(pkg.__name__ + '.fused')
cond = importlib.import_module(pkg.__name__ + '.conditioners')
WS = importlib.import_module(pkg.__name__ + '.workspace')
dev = torch.device('cuda:0')


def graph_us(fn, per_graph=20, replays=10):
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
    torch.manual_seed(0)
    mlp = cond.MLP(1, 2).to(dev).train()
    print('N       fwd multi-launch   fwd chain   [fwd+bwd multi   fwd+bwd chain]')
    for n in (256, 1024, 4096, 8192, 16384):
        x = torch.randn(n, 1, device=dev)
        gout = torch.randn(n, 2, device=dev)

        def old():
            with torch.no_grad():
                fused.mlp_forward(mlp, x, chain=False)

        def new():
            fused.mlp_chain_forward_nograd(mlp, x, True)

        def train(chain):
            def f():
                for p in mlp.parameters():
                    p.grad = None
                xx = x.detach().requires_grad_(True)
                y = fused.mlp_forward(mlp, xx, chain=chain)
                y.backward(gout)
            return f

        t_old, t_new = graph_us(old), graph_us(new)
        try:
            tt_old, tt_new = graph_us(train(False), 5), graph_us(train(True), 5)
        except Exception as ex:                                    # backward chain not wired yet
            tt_old = tt_new = float('nan')
            print('  (train timing skipped: %s)' % str(ex)[:80])
        print('%-7d %10.1f us %12.1f us %14.1f us %14.1f us' % (n, t_old, t_new, tt_old, tt_new))


if __name__ == '__main__':
    main()
