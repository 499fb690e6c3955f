"""Options database: the stand-in for ``petsc4py.init(sys.argv)`` + ``ts.setFromOptions()``.

The reference's drivers pass every argv token argparse does not know to PETSc
(``examples-pnode/ode_demo_petsc.py:46,63-66``) and ``ODEPetsc.setupTS`` ends with
``self.ts.setFromOptions()`` (``pnode/petsc_adjoint.py:775``), so PETSc command-line options
override the ``method`` keyword, the controller, tolerances and the checkpointing mode.  The
same spellings are honoured here: call ``pnode_amd.init(sys.argv)`` where the reference calls
``petsc4py.init(sys.argv)``; options can also come from the ``PETSC_OPTIONS`` environment
variable (PETSc reads it too) or ``set_option``.
"""
import os
import shlex

_DB = {}


def _is_key(tok):
    if not tok.startswith("-") or len(tok) < 2:
        return False
    try:
        float(tok)
        return False          # a negative number is a value, not a key
    except ValueError:
        return True


def parse(argv):
    """['-ts_adapt_type', 'none', '-ts_monitor'] -> {'ts_adapt_type': 'none', 'ts_monitor': ''}"""
    out = {}
    toks = list(argv)
    i = 0
    while i < len(toks):
        tok = toks[i]
        if _is_key(tok):
            key = tok.lstrip("-")
            if i + 1 < len(toks) and not _is_key(toks[i + 1]):
                out[key] = str(toks[i + 1])
                i += 2
            else:
                out[key] = ""
                i += 1
        else:
            i += 1
    return out


def init(argv=None):
    """Fill the database from PETSC_OPTIONS and `argv` (argv[0] is skipped like PETSc does)."""
    _DB.clear()
    env = os.environ.get("PETSC_OPTIONS")
    if env:
        _DB.update(parse(shlex.split(env)))
    if argv:
        _DB.update(parse(list(argv)[1:]))
    return dict(_DB)


def set_option(key, value=""):
    _DB[key.lstrip("-")] = "" if value is None else str(value)


def del_option(key):
    _DB.pop(key.lstrip("-"), None)


def clear():
    _DB.clear()


def get_all():
    return dict(_DB)


def truthy(value, default=True):
    """PETSc bool parsing: a bare flag means true."""
    if value is None:
        return default
    v = str(value).strip().lower()
    if v == "":
        return True
    return v in ("1", "true", "yes", "on")
