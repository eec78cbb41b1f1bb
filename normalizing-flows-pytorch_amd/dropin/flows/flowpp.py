"""flows.flowpp of the reference -> the engine's Flowpp."""
import importlib

Flowpp = importlib.import_module('normalizing-flows-pytorch_amd').Flowpp
