"""flows.iresblock of the reference -> the engine's invertible residual block (vector data)."""
import importlib

_pkg = importlib.import_module('normalizing-flows-pytorch_amd')
InvertibleResLinear, LipSwish, SpectralNorm = _pkg.InvertibleResLinear, _pkg.LipSwish, _pkg.SpectralNorm
