"""flows.glow of the reference -> the engine's Glow."""
import importlib

Glow = importlib.import_module('normalizing-flows-pytorch_amd').Glow
