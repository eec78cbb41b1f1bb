"""flows.maf of the reference -> the engine's MAF classes."""
import importlib

_pkg = importlib.import_module('normalizing-flows-pytorch_amd')
MADE, AutoregressiveTransfrom, MAF = _pkg.MADE, _pkg.AutoregressiveTransfrom, _pkg.MAF
