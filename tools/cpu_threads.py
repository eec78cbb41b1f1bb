"""cpu_baseline thread sweep: the oracle's full train step (same workload as bench.py's cpu_baseline leg) at several intra-op
thread counts on this box's host cores.  One measurement per round; bench.py keeps the fastest setting (`cpu_threads` of its CONFIGS).

    python tools/cpu_threads.py c4 8 16 32 64
"""
import os
import sys
import time
from types import SimpleNamespace as NS

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import importlib
    name = sys.argv[1]
    threads = [int(a) for a in sys.argv[2:]] or [8, 16, 32, 64]
    cfg = bench.CONFIGS[name]
    pkg = importlib.import_module(bench.PKG)
    nfdata = importlib.import_module(bench.PKG + '.data')
    from oracle import models as om
    from oracle import transforms as tf
    torch.manual_seed(0)
    np.random.seed(0)
    net = getattr(pkg, cfg['cls'])(cfg['dims'], cfg['datatype'], NS(layers=cfg['layers'], mixtures=cfg['mixtures']))
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    B = cfg['batch']
    y = nfdata.sample(cfg['data'], B, 1234)
    if cfg['data'] == 'cifar':
        y = y.reshape((B, ) + cfg['dims'])
    print('%s B=%d host cores %d' % (name, B, os.cpu_count()))
    for t in threads:
        torch.set_num_threads(t)
        sd = {k: v.clone() for k, v in sd0.items()}
        ora = om.FlowOracle(cfg['kind'], cfg['dims'], cfg['datatype'], cfg['layers'], sd, mixtures=cfg['mixtures'],
                            training=True).requires_grad_(True)
        opt = torch.optim.Adam(list(ora.parameters().values()), lr=1.0e-4)

        def step():
            opt.zero_grad()
            z, ld = ora.forward(y)
            loss = tf.nll_loss(z, ld)
            loss.backward()
            opt.step()
        step()
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < float(os.environ.get('SECONDS_PER', '8')) and n < 100:
            step()
            n += 1
        el = time.perf_counter() - t0
        print('threads %3d: %8.2f ms/step  %10.1f samples/s  (%d steps)' % (t, 1e3 * el / n, B * n / el, n), flush=True)


if __name__ == '__main__':
    main()
