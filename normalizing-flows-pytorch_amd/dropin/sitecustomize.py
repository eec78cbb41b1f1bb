"""
Start-up hook of the drop-in (imported by the interpreter's ``site`` module because ``dropin/`` is on PYTHONPATH).

``python main.py`` inserts the script's directory at ``sys.path[0]``, ahead of PYTHONPATH, so a plain path search would
find the reference's own ``flows`` package first.  This registers a meta-path finder that gives the top-level name
``flows`` to ``dropin/flows`` (whose ``__init__`` then appends the reference's directory to its ``__path__`` for the
modules the engine does not replace).  ``NF_DROPIN=0`` disables it.  Any other ``sitecustomize`` further along
``sys.path`` (the distribution's, say) is chained so that it still runs.
"""
import importlib.abc
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


class _FlowsFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != 'flows':
            return None
        pkg = os.path.join(_HERE, 'flows')
        return importlib.util.spec_from_file_location('flows', os.path.join(pkg, '__init__.py'), submodule_search_locations=[pkg])


def _install():
    if os.environ.get('NF_DROPIN', '1') == '0':
        return
    if not any(type(f).__name__ == '_FlowsFinder' for f in sys.meta_path):
        sys.meta_path.insert(0, _FlowsFinder())


def _chain():
    """Run the next sitecustomize on sys.path, which this file shadows."""
    for p in sys.path:
        cand = os.path.join(p or os.getcwd(), 'sitecustomize.py')
        if os.path.isfile(cand) and os.path.realpath(cand) != os.path.realpath(__file__):
            spec = importlib.util.spec_from_file_location('_nf_chained_sitecustomize', cand)
            mod = importlib.util.module_from_spec(spec)
            try:
                spec.loader.exec_module(mod)
            except Exception:  # a failing sitecustomize must not take the interpreter down
                pass
            return


_install()
_chain()
