"""per-launch time of the image Flow++ conditioner kernels (csrc/flowpp_img.hip) at the CIFAR shapes.
   python tools/probes/flowpp_img_kernels.py [batch]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pkg = importlib.import_module(bench.PKG)
N = importlib.import_module(bench.PKG + '._native')
dev = torch.device('cuda:0')
st = N.stream()


COLD = os.environ.get('NF_PROBE_COLD', '0') != '0'      # evict L2 / MALL between launches (a 512 MB fill): operands come from HBM, as
_flush = torch.empty(128 * 1024 * 1024, device=dev) if COLD else None     # in a real step, where another kernel wrote them


def timed(what, fn, reps=20):
    """device time per launch inside a hipGraph of back-to-back launches (no host launch cost in the number)"""
    if COLD:
        both = bench.graph_time_us(lambda: (_flush.fill_(1.0), fn()), dev, per_graph=10, replays=3)
        us = both - bench.graph_time_us(lambda: _flush.fill_(1.0), dev, per_graph=10, replays=3)
    else:
        us = bench.graph_time_us(fn, dev, per_graph=50, replays=4)
    print('    %-34s %8.1f us' % (what, us))
    return us


for I0, O, HW in ((6, 84, 16), (24, 336, 8), (96, 1344, 4)):
    print('I0 = %d, O = %d, %d x %d, B = %d' % (I0, O, HW, HW, B))
    r = lambda *s: torch.randn(*s, device=dev) * 0.3
    xin, x, a, x4, out = r(B, I0, HW, HW), r(B, 32, HW, HW), r(B, 32, HW, HW), r(B, 32, HW, HW), r(B, O, HW, HW)
    W0, b0, Wg, bg, W5, b5 = r(32, I0, 3, 3), r(32), r(32, 64, 3, 3), r(32), r(O, 32, 3, 3), r(O)
    l1g, l1b, pos, l2g, l2b = r(32, HW, HW), r(32, HW, HW), r(32, HW, HW), r(32, HW, HW), r(32, HW, HW)
    c1w, c1b, c2w, c2b = r(96, 32), r(96), r(64, 32), r(64)
    gcat = r(B, 64, HW, HW)
    gz = [torch.zeros_like(t) for t in (W0, b0, Wg, bg, l1g, l1b, pos, c1w, c1b, c2w, c2b, l2g, l2b, W5, b5)]
    p = N.ptr
    tot = 0.0
    tot += timed('conv0 fwd', lambda: N.call('nf_flowpp_img_conv', p(xin), p(W0), p(b0), p(x), B, I0, 32, HW, HW, 0, 0, 1, N.stream()))
    tot += timed('gated conv fwd', lambda: N.call('nf_flowpp_img_conv', p(x), p(Wg), p(bg), p(a), B, 64, 32, HW, HW, 1, 0, 1, N.stream()))
    tot += timed('mid fwd', lambda: N.call('nf_flowpp_img_mid_fwd', p(x), p(a), p(l1g), p(l1b), p(pos), p(c1w), p(c1b), p(c2w), p(c2b),
                                           p(l2g), p(l2b), p(x4), B, HW, HW, N.stream()))
    tot += timed('last conv fwd', lambda: N.call('nf_flowpp_img_conv', p(x4), p(W5), p(b5), p(out), B, 32, O, HW, HW, 0, 0, 1, N.stream()))
    print('    forward total %.1f us' % tot)
    tot = 0.0
    lib = N.load()
    ks = int(lib.nf_flowpp_img_conv_ksplit(B, O, 32, HW, HW))
    g4 = r(ks, B, 32, HW, HW)
    tot += timed('last conv dgrad (%d slabs)' % ks, lambda: N.call('nf_flowpp_img_conv', p(out), p(W5), None, p(g4), B, O, 32, HW, HW, 0, 1, ks, N.stream()))
    ns = int(lib.nf_flowpp_img_wgrad_slabs(B, 32, O, HW, HW)); sw = r(ns * O * 32 * 9); sb = r(ns * O)
    tot += timed('last conv wgrad (%d slabs)' % ns, lambda: N.call('nf_flowpp_img_conv_wgrad', p(x4), p(out), p(sw), p(sb), ns, B, 32, O, HW, HW, 0, N.stream()))
    tot += timed('mid bwd', lambda: N.call('nf_flowpp_img_mid_bwd', p(x), p(a), p(l1g), p(l1b), p(pos), p(c1w), p(c1b), p(c2w), p(c2b),
                                           p(l2g), p(l2b), p(g4), p(x4), p(out), p(gz[4]), p(gz[5]), p(gz[6]), p(gz[7]), p(gz[8]),
                                           p(gz[9]), p(gz[10]), p(gz[11]), p(gz[12]), 0, B, HW, HW, ks, N.stream()))
    tot += timed('gated conv dgrad', lambda: N.call('nf_flowpp_img_conv', p(a), p(Wg), None, p(gcat), B, 32, 64, HW, HW, 0, 1, 1, N.stream()))
    ns = int(lib.nf_flowpp_img_wgrad_slabs(B, 64, 32, HW, HW)); sw = r(ns * 32 * 64 * 9); sb = r(ns * 32)
    tot += timed('gated conv wgrad (%d slabs)' % ns, lambda: N.call('nf_flowpp_img_conv_wgrad', p(x), p(a), p(sw), p(sb), ns, B, 64, 32, HW, HW, 1, N.stream()))
    tot += timed('celu bwd', lambda: N.call('nf_flowpp_img_celu_bwd', p(x), p(gcat), p(x4), B, 32, HW, HW, N.stream()))
    ns = int(lib.nf_flowpp_img_wgrad_slabs(B, I0, 32, HW, HW)); sw = r(ns * 32 * I0 * 9); sb = r(ns * 32)
    tot += timed('conv0 wgrad (%d slabs)' % ns, lambda: N.call('nf_flowpp_img_conv_wgrad', p(xin), p(x), p(sw), p(sb), ns, B, I0, 32, HW, HW, 0, N.stream()))
    tot += timed('conv0 dgrad', lambda: N.call('nf_flowpp_img_conv', p(x), p(W0), None, p(xin), B, 32, I0, HW, HW, 0, 1, 1, N.stream()))
    print('    backward total %.1f us' % tot)
