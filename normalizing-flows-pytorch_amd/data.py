"""
Seeded synthetic inputs of the benchmark configurations.  The reference's data code (flows/dataset.py) needs
hydra / torchvision / sklearn tables and cannot travel; these are numpy restatements of the same distributions that
take the sample count ``n`` directly (flows/dataset.py:13-34; sklearn.datasets.make_moons / make_circles geometry).
"""
import numpy as np
import torch


def moons(n, rng, noise=0.08):
    """two interleaving half circles + gaussian noise, then (x - 0.5) / 2   (dataset.py:18-21)"""
    n_out = n // 2
    n_in = n - n_out
    t_out = np.linspace(0.0, np.pi, n_out)
    t_in = np.linspace(0.0, np.pi, n_in)
    x = np.concatenate([np.cos(t_out), 1.0 - np.cos(t_in)])
    y = np.concatenate([np.sin(t_out), 1.0 - np.sin(t_in) - 0.5])
    pts = np.stack([x, y], axis=1)
    pts = pts[rng.permutation(n)]
    pts = pts + rng.normal(scale=noise, size=pts.shape)
    return ((pts - 0.5) / 2.0).astype(np.float32)


def circles(n, rng, noise=0.08, factor=0.5):
    """two concentric circles (radii 1 and `factor`) + noise, scaled by 0.6   (dataset.py:13-15)"""
    n_out = n // 2
    n_in = n - n_out
    t_out = np.linspace(0.0, 2.0 * np.pi, n_out, endpoint=False)
    t_in = np.linspace(0.0, 2.0 * np.pi, n_in, endpoint=False)
    x = np.concatenate([np.cos(t_out), np.cos(t_in) * factor])
    y = np.concatenate([np.sin(t_out), np.sin(t_in) * factor])
    pts = np.stack([x, y], axis=1)
    pts = pts[rng.permutation(n)]
    pts = pts + rng.normal(scale=noise, size=pts.shape)
    return (pts * 0.6).astype(np.float32)


def normals(n, rng, radius=0.7, n_normals=8):
    """8 gaussians (sigma 0.1) on a circle of radius 0.7   (dataset.py:24-34)"""
    k = rng.integers(n_normals, size=(n, ))
    cx = radius * np.cos(2.0 * np.pi * k / n_normals)
    cy = radius * np.sin(2.0 * np.pi * k / n_normals)
    d = rng.normal(size=(2, n)) * 0.1
    return np.stack([cx + d[0], cy + d[1]], axis=1).astype(np.float32)


def cifar_like(n, rng, dims=(3, 32, 32)):
    """uniform uint8 pixels / 255 (the reference feeds uint8/255 without dequantisation noise, dataset.py:120)"""
    return (rng.integers(0, 256, size=(n, ) + tuple(dims), dtype=np.uint8).astype(np.float32) / 255.0)


GENERATORS = {'moons': moons, 'circles': circles, 'normals': normals, 'cifar': cifar_like}


def sample(name, n, seed):
    return torch.from_numpy(GENERATORS[name](n, np.random.default_rng(seed)))


# ---- on-device generation (csrc/datagen.hip): same distributions, drawn where they are consumed --------------------------------------
KINDS = {'moons': 0, 'circles': 1, 'normals': 2, 'cifar': 3}


class DeviceSampler:
    """A stream of synthetic batches generated ON the GPU: ``next()`` fills (and returns) one static tensor with a fresh batch.
    The step counter lives in device memory and the draw is a pure function of (seed, step, sample index), so the two launches
    can be captured into a hipGraph (FlowTrainer(sampler=...)): every replay trains on a new batch without any host-to-device
    copy -- the reference copies each batch from the host (main.py:79).

    ``rank``: under data parallelism every replica must draw a DIFFERENT shard of the global batch; the rank (default: the
    process group's, 0 without one) is folded into the Philox key, so W replicas built with the same seed see W distinct
    streams and the effective global batch is B * W."""

    def __init__(self, name, batch, dims, seed=0, device='cuda', rank=None):
        import torch as _t
        from . import _native as N
        self._N = N
        self.kind = KINDS[name]
        self.batch = int(batch)
        self.dims = tuple(dims)
        per = 1
        for d_ in self.dims:
            per *= int(d_)
        if self.kind != 3 and per != 2:
            raise ValueError('%s is a 2-D data set' % name)
        self.per = per
        if rank is None:
            import torch.distributed as _d
            rank = _d.get_rank() if _d.is_available() and _d.is_initialized() else 0
        self.rank = int(rank)
        self.seed = (int(seed) ^ (self.rank * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF    # 63-bit key, distinct per rank
        self.out = _t.empty((self.batch, ) + self.dims, dtype=_t.float32, device=device)
        self.step = _t.zeros(1, dtype=_t.int64, device=device)

    def next(self):
        N = self._N
        N.call('nf_sample_data', self.kind, self.out.data_ptr(), self.batch, self.per, self.seed, self.step.data_ptr(), N.stream())
        N.call('nf_sample_advance', self.step.data_ptr(), N.stream())
        return self.out
