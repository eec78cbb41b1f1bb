"""
On-device synthetic data (csrc/datagen.hip) against the host restatement of the reference's generators (data.py,
flows/dataset.py:13-34, :120): distribution-level agreement, reproducibility, and fresh batches under hipGraph replay.
Needs a real MI355X.
"""
import importlib
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _data(pkg):
    return importlib.import_module(pkg.__name__ + '.data')


@pytest.mark.parametrize('name', ['moons', 'circles', 'normals'])
def test_device_sampler_matches_host_distribution(pkg, name):
    D = _data(pkg)
    n = 1 << 18
    s = D.DeviceSampler(name, n, (2, ), seed=11, device=DEV)
    got = s.next().cpu().double().numpy()
    want = D.sample(name, n, 99).double().numpy()
    se = 4.0 / np.sqrt(n)                                    # ~4 standard errors of a unit-scale statistic
    assert np.abs(got.mean(0) - want.mean(0)).max() < se
    assert np.abs(np.cov(got.T) - np.cov(want.T)).max() < 2 * se
    # marginal quantiles and the radius distribution (separates the two moons / circles / the 8 modes' ring)
    qs = np.linspace(0.02, 0.98, 25)
    for a, b in ((got[:, 0], want[:, 0]), (got[:, 1], want[:, 1]), (np.hypot(*got.T), np.hypot(*want.T))):
        assert np.abs(np.quantile(a, qs) - np.quantile(b, qs)).max() < 0.01
    # 2-D histogram distance (total variation on a 24 x 24 grid)
    rng = [[-1.2, 1.2], [-1.2, 1.2]]
    h1 = np.histogram2d(got[:, 0], got[:, 1], bins=24, range=rng)[0] / n
    h2 = np.histogram2d(want[:, 0], want[:, 1], bins=24, range=rng)[0] / n
    assert 0.5 * np.abs(h1 - h2).sum() < 0.03


def test_device_sampler_cifar_like(pkg):
    D = _data(pkg)
    s = D.DeviceSampler('cifar', 256, (3, 32, 32), seed=5, device=DEV)
    x = s.next().cpu()
    assert x.shape == (256, 3, 32, 32)
    lev = x * 255.0
    assert float((lev - lev.round()).abs().max()) < 1e-4      # uint8 / 255 exactly
    assert float(x.min()) >= 0.0 and float(x.max()) <= 1.0
    hist = torch.bincount(lev.round().long().flatten(), minlength=256).double()
    assert float((hist / hist.sum() - 1.0 / 256).abs().max()) < 1.0e-3
    assert abs(float(x.mean()) - 0.5) < 2e-3


def test_device_sampler_is_reproducible_and_advances(pkg):
    D = _data(pkg)
    a = D.DeviceSampler('moons', 4096, (2, ), seed=3, device=DEV)
    b = D.DeviceSampler('moons', 4096, (2, ), seed=3, device=DEV)
    x0, y0 = a.next().clone(), b.next().clone()
    assert torch.equal(x0, y0)                               # pure function of (seed, step, index)
    x1 = a.next().clone()
    assert not torch.equal(x0, x1)                           # the device-side step moved on
    assert int(a.step.item()) == 2
    c = D.DeviceSampler('moons', 4096, (2, ), seed=4, device=DEV)
    assert not torch.equal(c.next(), x0)


def test_trainer_draws_a_fresh_batch_per_graph_replay(pkg):
    """FlowTrainer(sampler=...): the draw is captured into the step's hipGraph; replays see new data (the loss moves) and the
    device-side step counter counts them -- no host-to-device copy anywhere in the step."""
    D = _data(pkg)
    nftrain = importlib.import_module(pkg.__name__ + '.train')
    torch.manual_seed(0)
    net = pkg.Glow((2, ), '2d', NS(layers=4)).to(DEV)
    sampler = D.DeviceSampler('moons', 1024, (2, ), seed=1, device=DEV)
    trainer = nftrain.FlowTrainer(net, graph=True, warmup=2, sampler=sampler)
    losses, batches = [], []
    for _ in range(6):
        z, loss = trainer.train_on_batch()
        torch.cuda.synchronize()
        losses.append(float(loss))
        batches.append(sampler.out.clone())
    assert trainer._g_fb is not None
    assert int(sampler.step.item()) == int(trainer.optim.step_count.item())
    assert not torch.equal(batches[-1], batches[-2]) and not torch.equal(batches[-2], batches[-3])
    assert all(np.isfinite(losses)) and len(set(losses[-3:])) == 3
