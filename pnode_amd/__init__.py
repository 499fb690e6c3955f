"""pnode_amd -- MI355X-native explicit-RK neural-ODE time stepper + discrete-adjoint engine.

Drop-in for the ``ODEPetsc.setupTS / odeint / odeint_adjoint`` path of caidao22/pnode::

    import pnode_amd
    pnode_amd.init(sys.argv)            # where the reference calls petsc4py.init(sys.argv)
    from pnode_amd import petsc_adjoint # or: from pnode import petsc_adjoint (shim package)
    ode = petsc_adjoint.ODEPetsc()

See DESIGN.md for the path, the boundary and the kernels; INTEGRATION.md for switching over.
"""
import os as _os

# ROCm 7.2 (CLR "AQL packet capture" for hipGraph launches): a captured graph that contains PyTorch's
# two-pass reduction (e.g. the bias gradient of nn.Linear at batch 4096) replays with WRONG results
# after any hipStreamSynchronize/hipDeviceSynchronize.  Reproduced without any pnode_amd code
# (tools/graph_sum_repro2.py); DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 avoids it at no measurable cost
# (profiles/r01_graph_packet_capture.txt).  The flag is read when the HIP runtime initialises, so it
# is set here, at import; petsc_adjoint refuses to capture graphs when that was too late.
import torch as _torch

GRAPH_REPLAY_SAFE = (_os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"
                     or not _torch.cuda.is_initialized())
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
if _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] != "0":
    GRAPH_REPLAY_SAFE = False

from . import options  # noqa: F401,E402
from .options import init, set_option  # noqa: F401,E402
from . import petsc_adjoint  # noqa: F401,E402

__all__ = ["petsc_adjoint", "options", "init", "set_option"]
