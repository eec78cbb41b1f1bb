"""flows.resflow of the reference -> the engine's ResFlow."""
import importlib

ResFlow = importlib.import_module('normalizing-flows-pytorch_amd').ResFlow
