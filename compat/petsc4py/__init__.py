"""Opt-in stand-in for the three things the reference's drivers take from petsc4py, so that a driver written
for caidao22/pnode runs UNMODIFIED on pnode_amd:

    PYTHONPATH=<repo>/compat:<repo> python ode_demo_petsc.py -ts_adapt_type none ...

    import petsc4py; petsc4py.init(sys.argv)     -> fills pnode_amd's options database (same spellings)
    from petsc4py import PETSc                   -> a namespace with ScalarType (the reference's test asserts it)
    from pnode import petsc_adjoint              -> the shim package at the repository root

This directory is not on the path unless you put it there; it never shadows a real petsc4py by accident.
There is no PETSc behind it: anything else of petsc4py's API raises AttributeError.
"""
import types

import numpy as _np
import pnode_amd as _pnode_amd


def init(args=None, arch=None, comm=None):
    """petsc4py.init(sys.argv): hand the PETSc-style options to pnode_amd."""
    _pnode_amd.init(list(args) if args is not None else None)


def get_config():
    return {"PETSC_DIR": "", "PETSC_ARCH": "pnode_amd"}


PETSc = types.SimpleNamespace(
    ScalarType=_np.float64,       # states may be float32 or float64 at run time here (PETSc fixes one width per build)
    RealType=_np.float64,
    IntType=_np.int64,
)
__all__ = ["init", "get_config", "PETSc"]
