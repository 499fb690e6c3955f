"""``-log_view``: a summary of the solver's own device work at interpreter exit, in the spirit of PETSc's
option of the same name (which the reference's users get from petsc4py).  Every device entry point is then
launched with HIP start/stop events bound to its dispatch; sweeps are not captured into hipGraphs while the
events are on (they cannot be attached to graph nodes)."""
import atexit
import ctypes
import sys

from . import _lib

_state = {"on": False, "solves": 0, "backward": 0, "steps": 0, "rejections": 0, "nfe_f": 0, "nfe_b": 0}


def enable():
    if _state["on"]:
        return
    _state["on"] = True
    _lib.load().pn_prof_enable(1)
    atexit.register(report)


def note_forward(ode):
    if _state["on"]:
        _state["solves"] += 1
        _state["steps"] += int(ode.num_steps)
        _state["rejections"] += int(ode.num_rejections)


def note_backward(ode):
    if _state["on"]:
        _state["backward"] += 1


def report(file=None):
    file = file or sys.stdout
    lib = _lib.load()
    K = len(_lib.KERNEL_IDS)
    L, us, by = (ctypes.c_int64 * K)(), (ctypes.c_double * K)(), (ctypes.c_double * K)()
    if lib.pn_prof_collect(len(L), L, us, by):
        return
    print("-" * 96, file=file)
    print("pnode_amd -log_view: %d forward sweeps, %d reverse sweeps, %d accepted time steps, %d rejected attempts"
          % (_state["solves"], _state["backward"], _state["steps"], _state["rejections"]), file=file)
    print("%-22s %10s %12s %10s %12s   %s" % ("device entry point", "launches", "total ms", "avg us", "GB/s moved", "replaces (PETSc)"), file=file)
    what = {"pn_rk_stage": "VecCopy + VecMAXPY per stage", "pn_rk_combine_wrms": "TSEvaluateStep + TSErrorWeightedNorm",
            "pn_adj_theta": "VecMAXPY + VecScale (adjoint stage)", "pn_adj_accum": "closing VecMAXPY of TSAdjointStep",
            "pn_param_accum": "VecAXPY on the parameter sensitivities", "pn_copy": "VecCopy", "pn_dots": "VecMDot / VecNorm",
            "pn_lincomb": "VecMAXPY / VecWAXPY (implicit steppers)"}
    tot = 0.0
    for i, name in enumerate(_lib.KERNEL_IDS):
        if not L[i]:
            continue
        tot += us[i]
        print("%-22s %10d %12.3f %10.2f %12.1f   %s" % (name, L[i], us[i] / 1e3, us[i] / L[i], by[i] / us[i] / 1e3, what.get(name, "")), file=file)
    print("%-22s %10s %12.3f" % ("all", "", tot / 1e3), file=file)
    print("-" * 96, file=file)
