"""pnode_amd -- MI355X-native explicit-RK neural-ODE time stepper + discrete-adjoint engine.

Drop-in for the ``ODEPetsc.setupTS / odeint / odeint_adjoint`` path of caidao22/pnode::

    import pnode_amd
    pnode_amd.init(sys.argv)            # where the reference calls petsc4py.init(sys.argv)
    from pnode_amd import petsc_adjoint # or: from pnode import petsc_adjoint (shim package)
    ode = petsc_adjoint.ODEPetsc()

See DESIGN.md for the path, the boundary and the kernels; INTEGRATION.md for switching over.
"""
from . import options  # noqa: F401
from .options import init, set_option  # noqa: F401
from . import petsc_adjoint  # noqa: F401

__all__ = ["petsc_adjoint", "options", "init", "set_option"]
