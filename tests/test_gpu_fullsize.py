"""
Size-independent properties at the FULL sizes of the five BASELINE.json configs (oracle parity at these sizes:
tests/test_gpu_fullsize_parity.py):
  * round trip      backward(forward(y)) == y        (inverse flow undoes the forward flow)
  * log-det balance ld_forward + ld_inverse == 0
  * linearity of the index maps / determinism: two forward passes give bit-identical results
  * training step sanity: finite loss, every trainable parameter receives a finite gradient
Needs a real MI355X.
"""
import importlib
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'

CONFIGS = [
    # name, class, dims, datatype, layers, mixtures, batch, data, x tolerance, ld tolerance
    ('c1_realnvp_moons', 'RealNVP', (2, ), '2d', 32, None, 256, 'moons', 2e-4, 2e-3),
    ('c2_glow_moons', 'Glow', (2, ), '2d', 32, None, 4096, 'moons', 2e-4, 2e-3),
    ('c3_flowpp_circles', 'Flowpp', (2, ), '2d', 32, 8, 65536, 'circles', 5e-3, 5e-2),   # 32 bisection brackets stack up
    ('c4_glow_cifar', 'Glow', (3, 32, 32), 'image', 32, None, 64, 'cifar', 1e-2, 0.5),       # 161 steps, 3072 dims: fp32 drift
    ('c5_maf_normals', 'MAF', (2, ), '2d', 10, None, 16384, 'normals', 2e-4, 2e-3),
]


@pytest.mark.parametrize('cfg', CONFIGS, ids=[c[0] for c in CONFIGS])
def test_round_trip_and_training_step_at_full_size(pkg, cfg):
    name, cls, dims, datatype, layers, mix, B, data, tol_x, tol_ld = cfg
    nfdata = importlib.import_module(pkg.__name__ + '.data')
    train = importlib.import_module(pkg.__name__ + '.train')
    torch.manual_seed(0)
    np.random.seed(0)
    net = getattr(pkg, cls)(dims, datatype, NS(layers=layers, mixtures=mix)).to(DEV)
    y = nfdata.sample(data, B, 1234).to(DEV)
    if data == 'cifar':
        y = y.clamp(0.02, 0.98)                       # Logit(eps=0.01) clamps: round trips only hold inside the clamp

    # ---- one training step: loss finite, all grads finite (also performs the data-dependent ActNorm init) ----------
    net.train()
    z, ld = net(y)
    loss = train.nll_loss(z, ld)
    loss.backward()
    assert torch.isfinite(loss), name
    n_grad = 0
    for k, p in net.named_parameters():
        if p.requires_grad:
            assert p.grad is not None, k
            assert bool(torch.isfinite(p.grad).all()), k
            n_grad += 1
    assert n_grad >= 2 * layers
    bpd = train.bits_per_dim(float(loss.detach()), dims)
    assert 0.0 < bpd < 20.0, bpd

    # ---- evaluation mode: deterministic forward, round trip, log-det balance -------------------------------------------
    net.eval()
    with torch.no_grad():
        z1, ld1 = net(y)
        z2, ld2 = net(y)
        assert torch.equal(z1, z2) and torch.equal(ld1, ld2)
        assert z1.shape == y.shape and ld1.shape == (B, )
        x, ldi = net.backward(z1)
    err = float((x - y).abs().max())
    assert err <= tol_x, (name, err)
    bal = float((ld1 + ldi).abs().max())
    assert bal <= tol_ld * max(1.0, float(ld1.abs().max()) * 1e-3 + 1.0), (name, bal)
