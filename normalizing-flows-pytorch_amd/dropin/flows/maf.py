"""flows.maf of the reference -> the engine's MAF classes."""
import importlib

_pkg = importlib.import_module('normalizing-flows-pytorch_amd')
MADE, AutoregressiveTransfrom, MAF = _pkg.MADE, _pkg.AutoregressiveTransfrom, _pkg.MAF


def __getattr__(name):
    """Names the engine does not replace (helpers such as flows/maf.py's free functions) come from the reference checkout."""
    from . import reference_module
    try:
        return getattr(reference_module('maf'), name)
    except ImportError as e:
        raise AttributeError('flows.maf has no %r in the engine and no reference checkout is reachable (%s)' % (name, e))
