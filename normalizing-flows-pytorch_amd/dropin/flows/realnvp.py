"""flows.realnvp of the reference -> the engine's RealNVP."""
import importlib

RealNVP = importlib.import_module('normalizing-flows-pytorch_amd').RealNVP
