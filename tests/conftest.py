import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

PKG_NAME = 'normalizing-flows-pytorch_amd'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def pkg():
    """the product package (its directory name is not a Python identifier, so import it by string)."""
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope='session')
def ref_flows():
    """the live reference, importable only in the authoring container (never on the GPU box)."""
    from tests._ref import load_reference
    mod = load_reference()
    if mod is None:
        pytest.skip('/root/reference not present')
    return mod
