"""flows.coupling of the reference -> the engine's coupling layers."""
import importlib

_pkg = importlib.import_module('normalizing-flows-pytorch_amd')
AbstractCoupling, AffineCoupling, MixLogAttnCoupling = _pkg.AbstractCoupling, _pkg.AffineCoupling, _pkg.MixLogAttnCoupling
AdditiveCoupling = _pkg.AdditiveCoupling


def __getattr__(name):
    """Names the engine does not replace (helpers such as flows/coupling.py's free functions) come from the reference checkout."""
    from . import reference_module
    try:
        return getattr(reference_module('coupling'), name)
    except ImportError as e:
        raise AttributeError('flows.coupling has no %r in the engine and no reference checkout is reachable (%s)' % (name, e))
