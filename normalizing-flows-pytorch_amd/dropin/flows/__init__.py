"""
Drop-in shim: a package NAMED ``flows`` with the reference's module layout that re-exports the MI355X engine.

    PYTHONPATH=<repo>/normalizing-flows-pytorch_amd/dropin python main.py network=glow run.distrib=moons

``main.py`` of tatsy/normalizing-flows-pytorch imports (main.py:12-17)

    from flows import MAF, Glow, Ffjord, Flowpp, RealNVP, ResFlow, PlanarFlow
    from flows.misc import anomaly_hook
    from flows.dataset import FlowDataLoader
    from flows.modules import Logit, Identity

Resolution rules of this package:

* ``flows``, ``flows.modules``, ``flows.coupling``, ``flows.squeeze``, ``flows.maf``, ``flows.misc``, ``flows.glow``,
  ``flows.realnvp``, ``flows.flowpp``, ``flows.resflow``, ``flows.iresblock`` are THIS directory's files: the HIP-backed
  classes under the reference's names.
* Everything else the reference's package holds (``flows.dataset``, ``flows.spectral_norm``, ``flows.weight_norm`` ...)
  resolves to the USER'S OWN copy of the reference: its ``flows`` directory is appended to ``flows.__path__``.  It is
  found as ``$NF_REFERENCE_FLOWS`` or as the next ``flows`` package on ``sys.path`` (``python main.py`` puts the
  reference checkout at ``sys.path[0]``).  Nothing of the reference is copied or shipped.
* Families outside the accelerated hot path (``PlanarFlow``, ``Ffjord``; BASELINE.json north_star) run on the
  reference's own code, unmixed: the user's reference package is mounted a second time under the private name
  ``_nf_reference_flows`` (without executing its ``__init__``) and ``flows.planar / ffjord / cnf / odeint`` forward to
  it, so their relative imports (``from .modules import Compose, BatchNorm, deriv_tanh``, planar.py:6) see the
  reference's modules, not the engine's.  Without a reference checkout they raise ``NotImplementedError`` on use.

``python main.py`` puts the script's directory BEFORE ``PYTHONPATH`` on ``sys.path``, so the reference's own ``flows``
would win the plain path search; ``dropin/sitecustomize.py`` (imported by the interpreter at start-up because
``dropin/`` is on ``PYTHONPATH``) registers a finder that gives the name ``flows`` to this package.
"""
import importlib
import os
import sys
import types

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(_HERE)))
if _ROOT not in sys.path:
    sys.path.append(_ROOT)  # behind everything of the user: only the engine package is looked up there
_pkg = importlib.import_module('normalizing-flows-pytorch_amd')

REFERENCE_ALIAS = '_nf_reference_flows'


def _find_reference_flows():
    """Directory of the user's reference ``flows`` package, or None."""
    env = os.environ.get('NF_REFERENCE_FLOWS')
    if env:
        if not os.path.isfile(os.path.join(env, '__init__.py')):
            raise ImportError('NF_REFERENCE_FLOWS=%r is not a package directory (no __init__.py)' % env)
        return os.path.abspath(env)
    here = os.path.realpath(_HERE)
    for p in sys.path:
        cand = os.path.join(p or os.getcwd(), 'flows')
        if os.path.isfile(os.path.join(cand, '__init__.py')) and os.path.realpath(cand) != here:
            return os.path.abspath(cand)
    return None


REFERENCE_DIR = _find_reference_flows()
if REFERENCE_DIR is not None:
    __path__.append(REFERENCE_DIR)  # engine files first, the user's reference files for every other module name
    if REFERENCE_ALIAS not in sys.modules:
        _alias = types.ModuleType(REFERENCE_ALIAS)
        _alias.__path__ = [REFERENCE_DIR]
        _alias.__package__ = REFERENCE_ALIAS
        sys.modules[REFERENCE_ALIAS] = _alias


def reference_module(name):
    """``flows.<name>`` of the user's reference checkout, loaded under the private alias package."""
    if REFERENCE_DIR is None:
        raise ImportError('flows.%s is outside the MI355X hot path and is served from the reference checkout: put it on '
                          'sys.path after dropin/ (python main.py does) or set NF_REFERENCE_FLOWS=<reference>/flows' % name)
    return importlib.import_module(REFERENCE_ALIAS + '.' + name)


MAF, Glow, Flowpp, RealNVP, ResFlow = _pkg.MAF, _pkg.Glow, _pkg.Flowpp, _pkg.RealNVP, _pkg.ResFlow


def _outside(name, module):
    """The reference's own class when a checkout is reachable, else a class that raises on construction."""
    if REFERENCE_DIR is not None:
        try:
            return getattr(reference_module(module), name)
        except ImportError:  # a dependency of the reference's file is missing (e.g. no checkout of cnf.py)
            pass

    class _Missing:
        def __init__(self, *a, **k):
            raise NotImplementedError('%s is outside the MI355X hot path (BASELINE.json north_star); it runs on the reference '
                                      'implementation: make the reference checkout importable (NF_REFERENCE_FLOWS)' % name)
    _Missing.__name__ = name
    return _Missing


PlanarFlow, Ffjord = _outside('PlanarFlow', 'planar'), _outside('Ffjord', 'ffjord')

__all__ = ['PlanarFlow', 'RealNVP', 'Glow', 'Flowpp', 'MAF', 'ResFlow', 'Ffjord']
