"""flows.planar is outside the accelerated path: served by the user's reference checkout (see flows/__init__.py)."""
import sys

from . import reference_module

sys.modules[__name__] = reference_module('planar')
