"""
oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A CPU restatement (numpy for the integer index maps, torch-CPU fp32 for the
floating-point transforms) of the forward / inverse + log-det-Jacobian hot path
of tatsy/normalizing-flows-pytorch: flows/coupling.py, flows/squeeze.py,
flows/modules.py, flows/maf.py, flows/glow.py, flows/flowpp.py, flows/realnvp.py
(every function cites the reference file:line it follows).

Who may use it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` -- as the CHECKER / the reported CPU yard-stick only.  The
product package (``normalizing-flows-pytorch_amd/``) never imports it and has no
CPU fallback: its transforms raise if the HIP library is missing.

How it is pinned: the reference is pure Python and importable in the authoring
container, so (1) ``tests/test_oracle_vs_reference.py`` runs the oracle against
the live reference (skipped where /root/reference is absent, e.g. the GPU box)
and (2) ``tests/golden/*.npz`` hold input/output vectors captured from the
reference by ``tests/golden/make_goldens.py``; ``tests/test_oracle_golden.py``
checks the oracle against them everywhere.  The reference has no tests or
golden vectors of its own.

Design: purely functional over a ``state_dict`` with the reference's key names
(SURVEY.md section 8b), so the same oracle consumes the reference's weights
(pinning) and the product's weights (parity) unchanged.
"""
from . import indexmaps, transforms, nets, iresblock, models  # noqa: F401
