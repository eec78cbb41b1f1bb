#!/usr/bin/env python
"""
Run-to-run reproducibility of one train step (forward + NLL + backward into the flat bucket) from IDENTICAL state:

    python tools/determinism_probe.py [c1 c2 c3 c4 c5 rnvp_img fpp_img c4_b512 ...] [--runs 4] [--layers L]

After two eager steps (data-dependent initialisations done) the model's state_dict (buffers included: the flow BatchNorm heads centre
their sums at the running mean) is restored before each of RUNS repetitions of FlowTrainer._forward_backward on the same batch; z, the
loss and every gradient tensor are compared BITWISE with the first repetition.  Prints which tensors differ (and by how much) -- the
map from a parameter name to the kernel whose batch sums are ordered by float atomics.  NF_DETERMINISTIC=1 must print "bit-identical"
for every workload.
"""
import argparse
import importlib
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'normalizing-flows-pytorch_amd'


def main():
    import bench
    ap = argparse.ArgumentParser()
    ap.add_argument('configs', nargs='*', default=['c1', 'c2', 'c3', 'c4', 'c5'])
    ap.add_argument('--runs', type=int, default=4)
    ap.add_argument('--layers', type=int, default=None)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'determinism.txt'))
    args = ap.parse_args()
    pkg = importlib.import_module(PKG)
    nftrain = importlib.import_module(PKG + '.train')
    nfdata = importlib.import_module(PKG + '.data')
    dev = torch.device('cuda', 0)
    lines = ['NF_DETERMINISTIC=%s' % os.environ.get('NF_DETERMINISTIC', '0')]
    worst_any = False
    for name in args.configs:
        batch = None
        if name == 'c4_b512':
            name, batch = 'c4', 512
        cfg = dict(bench.CONFIGS[name])
        if args.layers:
            cfg['layers'] = args.layers
        B = batch or cfg['batch']
        torch.manual_seed(0)
        np.random.seed(0)
        net = getattr(pkg, cfg['cls'])(cfg['dims'], cfg['datatype'], NS(layers=cfg['layers'], mixtures=cfg['mixtures'])).to(dev)
        trainer = nftrain.FlowTrainer(net, graph=False)
        y = nfdata.sample(cfg['data'], B, 1234)
        if cfg['data'] == 'cifar':
            y = y.reshape((B, ) + cfg['dims'])
        y = y.to(dev)
        for _ in range(2):
            trainer.train_on_batch(y)
        torch.cuda.synchronize()
        sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
        names = [k for k, _ in net.named_parameters()]
        first = None
        diffs = {}
        for r in range(args.runs):
            net.load_state_dict(sd)
            z, loss = trainer._forward_backward(y)
            torch.cuda.synchronize()
            rec = {'z': z.detach().clone(), 'loss': loss.detach().clone()}
            for k, p in net.named_parameters():
                if p.grad is not None:
                    rec['grad/' + k] = p.grad.detach().clone()
            if first is None:
                first = rec
                continue
            for k, v in rec.items():
                if not torch.equal(v, first[k]):
                    d = float((v.double() - first[k].double()).abs().max())
                    s = float(first[k].double().abs().max())
                    diffs[k] = max(diffs.get(k, 0.0), d / max(s, 1e-30))
        tag = '%s (B = %d, layers = %d)' % (name, B, cfg['layers'])
        if not diffs:
            lines.append('%-40s bit-identical over %d runs (%d gradient tensors, z, loss)' % (tag, args.runs, len(first) - 2))
        else:
            worst_any = True
            lines.append('%-40s %d of %d quantities differ between runs; worst relative (to the tensor\'s largest entry):' % (tag, len(diffs), len(first)))
            # group by parameter kind (the name behind the last layer index)
            kinds = {}
            for k, v in diffs.items():
                parts = k.split('.')
                kind = '.'.join(p for p in parts if not p.isdigit())
                e = kinds.setdefault(kind, [0, 0.0, k])
                e[0] += 1
                if v >= e[1]:
                    e[1], e[2] = v, k
            for kind, (n, v, k) in sorted(kinds.items(), key=lambda t: -t[1][1]):
                lines.append('    %-70s x%-4d worst %.3e at %s' % (kind, n, v, k))
        print(lines[-1] if not diffs else '\n'.join(lines[-(len(kinds) + 1):]), flush=True)
        del trainer, net
        torch.cuda.empty_cache()
    lines.append('turnstile waits that gave up: %d; persistent-kernel timeouts: %d' % (pkg._native.deterministic_timeouts(), pkg._native.persistent_timeouts()))
    print(lines[-1])
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, 'a') as f:
        f.write('\n'.join(lines) + '\n')
    return 1 if worst_any else 0


if __name__ == '__main__':
    sys.exit(main())
