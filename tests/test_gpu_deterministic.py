"""
Deterministic mode (csrc/nf_det.h, include/nfhip.h:nf_deterministic, NF_DETERMINISTIC=1): the reference's CPU path reproduces itself bit
for bit and main.py:308-311 has a determinism switch; the engine's fast path orders some batch sums by float atomics (per-sample
log-dets of the slab kernels, per-channel statistics and parameter gradients of the layerwise kernels, a few folds).  In the mode every
such sum is added in a fixed order.

  * two runs of the train step of EVERY BASELINE.json config (and of the image RealNVP / Flow++ stacks and config 4's literal batch 512)
    from identical state are BIT-IDENTICAL: z, the loss and every gradient tensor;
  * the racing mode stays at rounding distance from the ordered one wherever its forward pass is order-free (C1 .. C5: z identical, the
    loss to a few ulp, the flat gradient to 1e-5 relative), i.e. what bench.py times computes the same numbers;
  * no turnstile wait gave up.
Needs a real MI355X.
"""
import importlib
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'

CASES = [
    # name, class, dims, datatype, layers, mixtures, batch, data
    ('c1_realnvp_moons', 'RealNVP', (2, ), '2d', 32, None, 256, 'moons'),
    ('c2_glow_moons', 'Glow', (2, ), '2d', 32, None, 4096, 'moons'),
    ('c3_flowpp_circles', 'Flowpp', (2, ), '2d', 32, 8, 65536, 'circles'),
    ('c4_glow_cifar', 'Glow', (3, 32, 32), 'image', 32, None, 64, 'cifar'),
    ('c5_maf_normals', 'MAF', (2, ), '2d', 10, None, 16384, 'normals'),
    ('c4_glow_cifar_b512', 'Glow', (3, 32, 32), 'image', 4, None, 512, 'cifar'),      # the large-batch convolution kernels (conv_bulk.hip)
    ('realnvp_cifar', 'RealNVP', (3, 32, 32), 'image', 4, None, 64, 'cifar'),         # flow-BatchNorm statistics of image data
    ('flowpp_cifar', 'Flowpp', (3, 32, 32), 'image', 2, 8, 64, 'cifar'),              # flowpp_img*.hip: per-sample workgroups, direct sinks
]
ORDER_FREE_FORWARD = ('c1_realnvp_moons', 'c2_glow_moons', 'c3_flowpp_circles', 'c4_glow_cifar', 'c5_maf_normals')


@pytest.fixture
def det_mode(pkg):
    """switches the mode on for the test and puts the previous setting back"""
    N = pkg._native
    was = N.deterministic()
    N.deterministic(True)
    yield N
    N.deterministic(was)


def _setup(pkg, case):
    name, cls, dims, datatype, layers, mix, B, data = case
    nfdata = importlib.import_module(pkg.__name__ + '.data')
    nftrain = importlib.import_module(pkg.__name__ + '.train')
    torch.manual_seed(0)
    np.random.seed(0)
    net = getattr(pkg, cls)(dims, datatype, NS(layers=layers, mixtures=mix)).to(DEV)
    trainer = nftrain.FlowTrainer(net, graph=False)
    y = nfdata.sample(data, B, 1234)
    if data == 'cifar':
        y = y.reshape((B, ) + dims)
    y = y.to(DEV)
    for _ in range(2):                                      # data-dependent initialisations done, fused launch paths reached
        trainer.train_on_batch(y)
    torch.cuda.synchronize()
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    return net, trainer, y, sd


def _run(net, trainer, y, sd):
    net.load_state_dict(sd)                                  # (buffers too: the flow-BatchNorm heads centre their sums at the running mean)
    z, loss = trainer._forward_backward(y)
    torch.cuda.synchronize()
    rec = {'z': z.detach().clone(), 'loss': loss.detach().clone()}
    for k, p in net.named_parameters():
        if p.grad is not None:
            rec['grad/' + k] = p.grad.detach().clone()
    return rec


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_train_step_is_bit_reproducible_in_deterministic_mode(pkg, det_mode, case):
    net, trainer, y, sd = _setup(pkg, case)
    first = _run(net, trainer, y, sd)
    assert len(first) > 20
    for rep in range(2):
        again = _run(net, trainer, y, sd)
        differ = [k for k, v in again.items() if not torch.equal(v, first[k])]
        assert not differ, '%s: %d of %d quantities differ between two runs from identical state, e.g. %s' % (case[0], len(differ), len(first), differ[:5])
    assert det_mode.deterministic_timeouts() == 0
    assert det_mode.persistent_timeouts() == 0


@pytest.mark.parametrize('case', [c for c in CASES if c[0] in ORDER_FREE_FORWARD], ids=[c[0] for c in CASES if c[0] in ORDER_FREE_FORWARD])
def test_racing_mode_is_at_rounding_distance_from_the_ordered_mode(pkg, case):
    """what bench.py times (atomics racing) against the ordered mode from the same state: the forward pass of the five BASELINE configs has
    no racing sum that feeds z (identical bits); the loss differs by the order of one block-level sum, the gradients by the order of
    the sums the mode serialises -- rounding, not a different computation."""
    N = pkg._native
    was = N.deterministic()
    try:
        N.deterministic(False)
        net, trainer, y, sd = _setup(pkg, case)
        fast = _run(net, trainer, y, sd)
        N.deterministic(True)
        slow = _run(net, trainer, y, sd)
    finally:
        N.deterministic(was)
    assert torch.equal(fast['z'], slow['z']), 'z differs between the racing and the ordered mode'
    assert abs(float(fast['loss']) - float(slow['loss'])) <= 4 * 1.2e-7 * max(1.0, abs(float(slow['loss'])))
    num = den = 0.0
    for k, v in slow.items():
        if k.startswith('grad/'):
            d = (fast[k].double() - v.double()).reshape(-1)
            num += float(d @ d)
            den += float(v.double().reshape(-1) @ v.double().reshape(-1))
    rel = (num / max(den, 1e-300)) ** 0.5
    assert rel <= 1.0e-5, 'flat gradient: racing vs ordered mode %.3e relative' % rel
    assert N.deterministic_timeouts() == 0
