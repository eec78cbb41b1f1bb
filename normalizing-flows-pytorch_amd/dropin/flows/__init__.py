"""
Drop-in shim: a package NAMED ``flows`` with the reference's module layout that re-exports the MI355X engine.

    PYTHONPATH=<repo>/normalizing-flows-pytorch_amd/dropin:$PYTHONPATH python main.py network=glow run.distrib=moons

``main.py`` of tatsy/normalizing-flows-pytorch imports ``from flows import MAF, Glow, Flowpp, RealNVP, ...`` and
``from flows.modules import Logit, Identity`` (main.py:12-17); with this directory ahead of the reference on
``sys.path`` those names resolve to the HIP-backed classes, everything else in main.py stays untouched.
Families outside the accelerated hot path (PlanarFlow, Ffjord) are not provided here: import them from the
reference package under another name if needed.
"""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_pkg = importlib.import_module('normalizing-flows-pytorch_amd')

MAF, Glow, Flowpp, RealNVP, ResFlow = _pkg.MAF, _pkg.Glow, _pkg.Flowpp, _pkg.RealNVP, _pkg.ResFlow


def _missing(name):
    class _Missing:
        def __init__(self, *a, **k):
            raise NotImplementedError('%s is outside the MI355X hot path (BASELINE.json north_star); use the reference '
                                      'implementation for it' % name)
    _Missing.__name__ = name
    return _Missing


PlanarFlow, Ffjord = _missing('PlanarFlow'), _missing('Ffjord')

__all__ = ['PlanarFlow', 'RealNVP', 'Glow', 'Flowpp', 'MAF', 'ResFlow', 'Ffjord']
