"""Access to the committed golden fixtures (tests/golden/*.npz, produced by tests/golden/make_goldens.py)."""
import functools
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@functools.lru_cache(maxsize=None)
def _load(name):
    with np.load(os.path.join(GOLDEN_DIR, name + '.npz')) as f:
        return {k: f[k] for k in f.files}


def group(name, prefix, device='cpu'):
    """all arrays of ``name``.npz whose key starts with ``prefix`` (prefix stripped) as torch tensors."""
    out = {}
    for k, v in _load(name).items():
        if k.startswith(prefix):
            t = torch.from_numpy(v.copy())
            out[k[len(prefix):]] = t.to(device)
    if not out:
        raise KeyError('%s: no keys under %r' % (name, prefix))
    return out


def keys(name):
    return list(_load(name).keys())


MODEL_CASES = {
    # name: (oracle kind, product class, dims, datatype, layers, mixtures)
    'realnvp2d': ('realnvp', 'RealNVP', (2, ), '2d', 2, None),
    'glow2d': ('glow', 'Glow', (2, ), '2d', 2, None),
    'flowpp2d': ('flowpp', 'Flowpp', (2, ), '2d', 2, 8),
    'maf2d': ('maf', 'MAF', (2, ), '2d', 2, None),
    'glow_img': ('glow', 'Glow', (3, 16, 16), 'image', 1, None),
    'resflow2d': ('resflow', 'ResFlow', (2, ), '2d', 2, None),
    'realnvp_img': ('realnvp', 'RealNVP', (3, 16, 16), 'image', 1, None),
    'flowpp_img': ('flowpp', 'Flowpp', (3, 16, 16), 'image', 1, 4),
}


def seed_noise(s):
    """the seeds make_goldens.py sets before each stochastic pass (train forward 777, train inverse 778, eval 779)."""
    import numpy as np
    torch.manual_seed(s)
    np.random.seed(s)


def assert_close(a, b, atol, rtol=1e-5, what=''):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    bound = atol + rtol * b.abs()
    assert bool((err <= bound).all()), '%s: max abs err %.3e (atol %.1e rtol %.1e)' % (what, float(err.max()), atol, rtol)
