"""pytest plugin (-p pnode_amd_ref_plugin) used by tests/test_reference_suite.py: makes the reference's own test
file importable against pnode_amd in this GPU-less container -- <repo>/compat/petsc4py stands in for petsc4py,
the `pnode` shim package for the reference's package, and the test-suite's CPU stand-in for the six device entry
points is injected as backend (the product itself refuses CPU tensors).  Nothing of the reference is copied."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "compat"), ROOT, os.path.join(ROOT, "tests")]
sys.dont_write_bytecode = True

# Logging packages some of the reference's drivers import and this image lacks: a writer that accepts the calls and
# drops them (the drivers only log scalars through it).
if "tensorboardX" not in sys.modules:
    try:
        import tensorboardX  # noqa: F401
    except ImportError:
        import types

        class _NullWriter(object):
            def __init__(self, *a, **k):
                pass

            def __getattr__(self, name):
                return lambda *a, **k: None

        _tbx = types.ModuleType("tensorboardX")
        _tbx.SummaryWriter = _NullWriter
        sys.modules["tensorboardX"] = _tbx

# torchvision / torchsummary (train-Cifar10.py): synthetic CIFAR-shaped data and no-op helpers from ref_harness/stubs,
# only when the real packages are absent
for _name in ("torchvision", "torchsummary"):
    try:
        __import__(_name)
    except ImportError:
        _stubs = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")
        if _stubs not in sys.path:
            sys.path.append(_stubs)

from pnode_amd import petsc_adjoint  # noqa: E402
from _cpu_vecops import CpuVecOps  # noqa: E402

_orig_init = petsc_adjoint.ODEPetsc.__init__


def _init_with_cpu_stand_in(self, backend=None):
    _orig_init(self, backend=CpuVecOps)


petsc_adjoint.ODEPetsc.__init__ = _init_with_cpu_stand_in
