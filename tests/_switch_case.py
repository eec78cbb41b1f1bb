"""One train-mode forward + backward of a small model, results to an .npz -- run by tests/test_gpu_switches.py once in-process (defaults) and
once per environment switch in a subprocess with the switch at its non-default value.

    python tests/_switch_case.py <case> <out.npz>
"""
import importlib
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CASES = {
    # case: (golden model name or None, class, dims, datatype, layers, mixtures, batch)
    'glow2d': ('glow2d', 'Glow', (2, ), '2d', 2, None, 64),
    'realnvp2d': ('realnvp2d', 'RealNVP', (2, ), '2d', 2, None, 64),
    'maf2d': ('maf2d', 'MAF', (2, ), '2d', 2, None, 64),
    'glow_img': ('glow_img', 'Glow', (3, 16, 16), 'image', 1, None, 4),
    'flowpp_img': ('flowpp_img', 'Flowpp', (3, 16, 16), 'image', 1, 4, 4),
    'glow_img_b320': (None, 'Glow', (3, 16, 16), 'image', 1, None, 320),      # 8 x 8 maps x 320 samples: beyond the persistent chain
}


def run(case):
    from tests import _golden as G
    pkg = importlib.import_module('normalizing-flows-pytorch_amd')
    nftrain = importlib.import_module('normalizing-flows-pytorch_amd.train')
    gold, cls, dims, datatype, layers, mix, B = CASES[case]
    torch.manual_seed(100)
    np.random.seed(100)
    net = getattr(pkg, cls)(dims, datatype, NS(layers=layers, mixtures=mix, logdet='exact', spnorm_coeff=0.9))
    if gold is not None:
        net.load_state_dict(G.group('model_' + gold, 'sd0/'), strict=True)
        y = G.group('model_' + gold, '')['y']
    else:
        y = torch.rand((B, ) + dims, generator=torch.Generator().manual_seed(101))
    net = net.to('cuda').train()
    out = {}
    # two steps through the trainer's launch path (flat bucket, deferred queues): the second one is past the data-dependent initialisations
    trainer = nftrain.FlowTrainer(net, graph=False)
    yd = y.to('cuda')
    for step in range(2):
        np.random.seed(7)                                   # (MADE draws its masks from the global numpy stream)
        z, loss = trainer._forward_backward(yd)
        torch.cuda.synchronize()
        out['step%d/z' % step] = z.detach().cpu().numpy()
        out['step%d/loss' % step] = loss.detach().cpu().numpy()
        for k, p in net.named_parameters():
            if p.grad is not None:
                out['step%d/grad/%s' % (step, k)] = p.grad.detach().cpu().numpy().copy()
        trainer.optim.step()
    assert pkg._native.persistent_timeouts() == 0
    return out


if __name__ == '__main__':
    np.savez(sys.argv[2], **run(sys.argv[1]))
