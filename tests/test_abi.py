"""The C-ABI shared library loads on a CPU-only box and exports exactly what
include/pnode_amd.h declares; the ctypes binding mirrors the header.  No device calls here."""
import ctypes
import os
import re

import pytest

from oracle import ts_oracle
from pnode_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "pnode_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pn_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "declared in pnode_amd.h but not exported: " + n
    assert sorted(_lib.PROTOTYPES) == names, set(names) ^ set(_lib.PROTOTYPES)


def test_every_header_entry_cites_the_reference_interface_it_replaces():
    src = open(os.path.join(ROOT, "include", "pnode_amd.h")).read()
    assert src.count("pa.py:") >= 12


def test_abi_version_and_error_channel():
    lib = _lib.load()
    assert lib.pn_abi_version() == 3
    t = _lib.Tableau()
    assert lib.pn_tableau_get(b"no-such-tableau", ctypes.byref(t)) != 0
    assert b"no-such-tableau" in lib.pn_last_error()
    with pytest.raises(_lib.PnError):
        _lib.check(lib.pn_tableau_get(b"nope", ctypes.byref(t)))


@pytest.mark.parametrize("name", ["1fe", "midpoint", "2a", "2b", "3", "3bs", "4", "5f", "5dp"])
def test_tableaus_equal_the_oracles(name):
    a = _lib.get_tableau(name)
    b = ts_oracle.tableau_info(name)
    assert (a.s, a.order, bool(a.fsal), bool(a.has_embed)) == (b["s"], b["order"], b["fsal"], b["has_embed"])
    for i in range(a.s):
        assert a.b[i] == b["b"][i] and a.bembed[i] == b["bembed"][i]
        assert a.c[i] == pytest.approx(b["c"][i], abs=1e-15)
        for j in range(a.s):
            assert a.A[i][j] == b["A"][i, j]


def test_method_map_follows_the_reference():
    """pa.py:641-650 plus the midpoint extension; unknown names fall to PETSc's default 3bs."""
    lib = _lib.load()
    for m, rk in ts_oracle.METHOD_TO_RK.items():
        assert lib.pn_method_to_rk_type(m.encode()).decode() == rk
    for m in ["rk3", "dopri5_fixed", "adams", ""]:
        assert lib.pn_method_to_rk_type(m.encode()).decode() == ts_oracle.PETSC_DEFAULT_RK


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libpnode_amd.so"))
    with pytest.raises(ImportError, match="no fallback"):
        _lib.load()


def test_product_has_no_cpu_path_and_never_imports_the_oracle():
    import torch
    from pnode_amd import petsc_adjoint
    ode = petsc_adjoint.ODEPetsc()
    with pytest.raises(RuntimeError, match="no CPU path"):
        ode.setupTS(torch.zeros(4, 2), torch.nn.Linear(2, 2), method="rk4")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pnode_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "libpnoracle" not in text, f


def test_no_kernel_of_the_code_object_uses_scratch_memory(tmp_path):
    """Every gfx950 kernel of libpnode_amd.so keeps its working set in registers: private segment 0, no
    VGPR spills (SGPRs spilt into VGPR lanes touch no memory and are allowed).  (Round 2: a helper that took the 3 KiB argument block of pn_param_accum_multi by
    reference made the compiler copy it to scratch -- 40x slower, same results; only this check sees that.)"""
    import re
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "clang-offload-bundler")):
        pytest.skip("ROCm LLVM tools not installed")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "co.o")
    subprocess.run([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", _lib.LIB_PATH, fat], check=True)
    subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co], check=True)
    notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    kernels = re.findall(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?"
                         r"\.vgpr_spill_count:\s+(\d+)", notes, flags=re.S)
    assert len(kernels) > 100 and any("pn_param_accum_multi_kernel" in k[0] for k in kernels)
    bad = [k for k in kernels if int(k[1]) or int(k[3])]
    assert not bad, bad[:5]


def test_step_loop_entry_points_refuse_null_arguments_and_report_a_failed_callback():
    """include/pnode_amd.h section 3a without a device: the argument checks of pn_rk_attempt / pn_rk_adjoint_step, and the
    error path of a callback that fails (returns 0 / -1) -- with a pn_vec_ops table of host functions, as the CPU-only
    test container drives the loops."""
    lib = _lib.load()
    ts = ctypes.c_void_p(lib.pn_ts_create())
    _lib.check(lib.pn_ts_set_rk_type(ts, b"4"))
    cb = _lib.STAGE_CB(lambda user, stage, t: 0)                       # "evaluation failed"
    vcb = _lib.VJP_CB(lambda user, stage, t, in_w, scale: -1)
    kout = (ctypes.c_void_p * _lib.PN_MAX_STAGES)()
    ys = (ctypes.c_void_p * _lib.PN_MAX_STAGES)()
    assert lib.pn_rk_attempt(None, _lib.PN_F64, 4, ts, None, 0.0, 0.1, None, None, ys, None, 0, 0.0, cb, None, 0, None, None, kout) != 0
    assert b"null argument" in lib.pn_last_error()
    assert lib.pn_rk_adjoint_step(None, _lib.PN_F64, 4, ts, None, 0.0, 0.1, None, None, vcb, None, None) != 0
    assert b"null argument" in lib.pn_last_error()
    # host functions in the table: nothing is launched on a device
    calls = []
    noop = lambda *a: calls.append(a[0:1]) or 0
    ops = _lib.VecOps(_lib.RK_STAGE_FN(noop), _lib.RK_COMBINE_WRMS_FN(noop), _lib.ADJ_THETA_FN(noop), _lib.ADJ_ACCUM_FN(noop))
    buf = (ctypes.c_double * 4)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    for i in range(_lib.PN_MAX_STAGES):
        ys[i] = p
    rc = lib.pn_rk_attempt(None, _lib.PN_F64, 4, ts, ctypes.byref(ops), 0.0, 0.1, p, p, ys, None, 0, 0.0, cb, None, 0, None, None, kout)
    assert rc != 0 and b"stage callback failed" in lib.pn_last_error() and not calls      # stage 0 is evaluated before any launch
    rc = lib.pn_rk_adjoint_step(None, _lib.PN_F64, 4, ts, ctypes.byref(ops), 0.0, 0.1, p, p, vcb, None, None)
    assert rc != 0 and b"VJP callback failed" in lib.pn_last_error()
    # and a successful walk: rk4 = 3 stage launches + the closing combination, 4 evaluations at t + c_i h
    seen = []
    ok = _lib.STAGE_CB(lambda user, stage, t: seen.append((stage, t)) or p.value)
    _lib.check(lib.pn_rk_attempt(None, _lib.PN_F64, 4, ts, ctypes.byref(ops), 1.0, 0.5, p, p, ys, None, 0, 0.0, ok, None, 0, None, None, kout))
    assert seen == [(0, 1.0), (1, 1.25), (2, 1.25), (3, 1.5)] and len(calls) == 4
    lib.pn_ts_destroy(ts)
