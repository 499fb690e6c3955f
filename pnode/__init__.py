"""Import-path shim: every caller of the reference does ``from pnode import petsc_adjoint``
(e.g. ``examples-pnode/ode_demo_petsc.py:73``); this keeps that line working on top of
``pnode_amd``.  Nothing else lives here."""
from pnode_amd import init, petsc_adjoint  # noqa: F401  (init(argv) stands where petsc4py.init(argv) did)
