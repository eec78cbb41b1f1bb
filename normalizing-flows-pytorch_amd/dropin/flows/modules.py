"""flows.modules of the reference -> the engine's layers and conditioners."""
import importlib

_pkg = importlib.import_module('normalizing-flows-pytorch_amd')
_cond = importlib.import_module('normalizing-flows-pytorch_amd.conditioners')

Identity, Logit, ActNorm, BatchNorm, Compose, InvertibleConv1x1 = (_pkg.Identity, _pkg.Logit, _pkg.ActNorm, _pkg.BatchNorm,
                                                                   _pkg.Compose, _pkg.InvertibleConv1x1)
MLP, ConvNet, ResBlockLinear, ResBlock2d = _cond.MLP, _cond.ConvNet, _cond.ResBlockLinear, _cond.ResBlock2d
GatedLinear, GatedConv2d, GatedAttn, WeightNorm = _cond.GatedLinear, _cond.GatedConv2d, _cond.GatedAttn, _cond.WeightNorm
MixLogCDF = _pkg.MixLogCDF
Sigmoid, Tanh, Arctanh = _pkg.Sigmoid, _pkg.Tanh, _pkg.Arctanh


def __getattr__(name):
    """Names the engine does not replace (helpers such as flows/modules.py's free functions) come from the reference checkout."""
    from . import reference_module
    try:
        return getattr(reference_module('modules'), name)
    except ImportError as e:
        raise AttributeError('flows.modules has no %r in the engine and no reference checkout is reachable (%s)' % (name, e))
