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
    # The oracle is tiny-op torch-CPU work: on the GPU box's 256 host cores the default intra-op thread count is far past the optimum
    # (tools/cpu_threads.py, profiles/r04_cpu_threads.txt: the C4 oracle step takes 1.2 s at 16 threads, 2.7 s at 32, 7.0 s at 64; C1 gets
    # slower with every thread added) and the full-size parity tests spend their time there.  Cap it (NF_TEST_CPU_THREADS overrides).
    try:
        import torch
        cap = int(os.environ.get('NF_TEST_CPU_THREADS', '16'))
        if cap > 0 and torch.get_num_threads() > cap:
            torch.set_num_threads(cap)
    except Exception:
        pass


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
