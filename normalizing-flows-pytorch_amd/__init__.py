"""
MI355X-native flow-transform engine: the forward / inverse + log-det-Jacobian hot path of
tatsy/normalizing-flows-pytorch as hand-written HIP kernels for gfx950 (libnfhip.so, C ABI in include/nfhip.h)
behind the reference's nn.Module surface.  See DESIGN.md and INTEGRATION.md at the repository root.

The directory name is not a Python identifier; import it with
``importlib.import_module('normalizing-flows-pytorch_amd')`` or put ``<this dir>/dropin`` on ``sys.path`` and
``import flows`` (the drop-in shim with the reference's module names).
"""
from . import _native, functional  # noqa: F401
from ._build import build  # noqa: F401
from .layers import (MADE, ActNorm, AbstractCoupling, AdditiveCoupling, AffineCoupling, AutoregressiveTransfrom, BatchNorm, Compose, Identity,
                     InvertibleConv1x1, Logit, MixLogAttnCoupling, MixLogCDF, Squeeze2d, Unsqueeze2d, Sigmoid, Tanh, Arctanh, Squeeze1d,
                     Unsqueeze1d)
from .inverse_grad import differentiable_inverse  # noqa: F401
from .models import MAF, Flowpp, Glow, RealNVP
from .resflow import InvertibleResLinear, LipSwish, ResFlow, SpectralNorm

__all__ = ['ActNorm', 'AbstractCoupling', 'AdditiveCoupling', 'AffineCoupling', 'BatchNorm', 'Compose', 'Identity', 'InvertibleConv1x1',
           'Logit', 'Squeeze2d', 'Unsqueeze2d', 'Glow', 'RealNVP', 'Flowpp', 'MAF', 'MADE', 'AutoregressiveTransfrom',
           'MixLogAttnCoupling', 'MixLogCDF', 'Sigmoid', 'Tanh', 'Arctanh', 'Squeeze1d', 'Unsqueeze1d', 'ResFlow', 'InvertibleResLinear', 'SpectralNorm', 'LipSwish', 'build', 'functional', 'differentiable_inverse']
