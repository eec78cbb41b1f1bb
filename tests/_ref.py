"""Loads the upstream reference (``/root/reference/flows``) under the alias ``ref_flows`` so that it never
collides with anything of ours.  Used ONLY by tests that pin the oracle and by tests/golden/make_goldens.py,
and only in the authoring container: the reference does not exist on the GPU box."""
import importlib.util
import os
import sys
import warnings

REF_ROOT = os.environ.get('NF_REFERENCE_ROOT', '/root/reference')


def load_reference(alias='ref_flows'):
    if alias in sys.modules:
        return sys.modules[alias]
    init = os.path.join(REF_ROOT, 'flows', '__init__.py')
    if not os.path.exists(init):
        return None
    sys.dont_write_bytecode = True          # /root/reference is read-only
    warnings.filterwarnings('ignore', message='torch.lu is deprecated')
    warnings.filterwarnings('ignore', message='torch.lu_solve is deprecated')
    spec = importlib.util.spec_from_file_location(alias, init,
                                                  submodule_search_locations=[os.path.join(REF_ROOT, 'flows')])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[alias] = mod
    spec.loader.exec_module(mod)
    return mod
