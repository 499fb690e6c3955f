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
    header = open(os.path.join(ROOT, "include", "pnode_amd.h")).read()
    declared = int(re.search(r"#define PN_ABI_VERSION (\d+)", header).group(1))
    assert lib.pn_abi_version() == declared == _lib.PN_ABI_VERSION == 4
    assert int(re.search(r"#define PN_WGRAD_MAX_PAIRS (\d+)", header).group(1)) == _lib.PN_WGRAD_MAX_PAIRS
    # the kernel-id enum is versioned with the ABI: its length is what the binding's name table has
    enum = re.search(r"typedef enum \{([^}]*)\} pn_kernel_id;", header).group(1)
    ids = [x.strip().split("=")[0].strip() for x in enum.split(",")]
    assert ids[-1] == "PN_K_COUNT" and len(ids) - 1 == len(_lib.KERNEL_IDS)
    for k, name in enumerate(_lib.KERNEL_IDS):
        assert lib.pn_kernel_name(k).decode() == name
    t = _lib.Tableau()
    assert lib.pn_tableau_get(b"no-such-tableau", ctypes.byref(t)) != 0
    assert b"no-such-tableau" in lib.pn_last_error()
    with pytest.raises(_lib.PnError):
        _lib.check(lib.pn_tableau_get(b"nope", ctypes.byref(t)))


def test_prof_collect_fills_no_more_than_the_callers_arrays_hold():
    """pn_prof_collect(count, ...): a client built against a shorter pn_kernel_id enum passes its own array length and is not
    overrun (PN_K_COUNT grew from 8 to 9 under an unchanged ABI version)."""
    lib = _lib.load()
    n = len(_lib.KERNEL_IDS)
    for count in (0, 3, n, n + 4):
        L = (ctypes.c_int64 * (n + 8))(*([-7] * (n + 8)))
        us = (ctypes.c_double * (n + 8))(*([-7.0] * (n + 8)))
        by = (ctypes.c_double * (n + 8))(*([-7.0] * (n + 8)))
        assert lib.pn_prof_collect(count, L, us, by) == 0
        filled = min(count, n)
        assert all(L[i] == 0 and us[i] == 0.0 and by[i] == 0.0 for i in range(filled))
        assert all(L[i] == -7 and us[i] == -7.0 and by[i] == -7.0 for i in range(filled, n + 8))


def test_the_wgrad_pair_struct_matches_the_header_layout():
    """pn_wgrad_pair as ctypes sees it: four pointers, a double, two int64 -- 56 bytes, no padding; the argument checks of
    pn_linear_wgrad_group run before any launch."""
    assert ctypes.sizeof(_lib.WgradPair) == 56
    lib = _lib.load()
    arr = (_lib.WgradPair * 1)()
    assert lib.pn_linear_wgrad_group(None, _lib.PN_F32, 256, 0, arr, 0) != 0 and b"npairs" in lib.pn_last_error()
    assert lib.pn_linear_wgrad_group(None, _lib.PN_F32, 256, 9, arr, 0) != 0
    arr[0].out_f, arr[0].in_f = 64, 60
    assert lib.pn_linear_wgrad_group(None, _lib.PN_F32, 256, 1, arr, 0) != 0 and b"unsupported" in lib.pn_last_error()
    arr[0].out_f, arr[0].in_f, arr[0].g, arr[0].x, arr[0].pw = 64, 64, 16, 32, 8
    assert lib.pn_linear_wgrad_group(None, _lib.PN_F32, 256, 1, arr, 0) != 0 and b"aligned" in lib.pn_last_error()
    assert lib.pn_linear_wgrad_supported(_lib.PN_F64, 256, 64, 64) == 1 and lib.pn_linear_wgrad_supported(_lib.PN_F64, 255, 64, 64) == 0
    assert lib.pn_linear_wgrad_supported(_lib.PN_F32, 1000, 64, 64) == 1            # ragged row counts: zero-filled tail
    nb = ctypes.c_int64()
    assert lib.pn_linear_wgrad_work_bytes(_lib.PN_F32, 512, 512, ctypes.byref(nb)) == 8 * 512 * 512 * 4 and nb.value == 8 * 8 * 512 * 8
    assert lib.pn_linear_wgrad_work_bytes(_lib.PN_F64, 512, 512, None) == 8 * 512 * 512 * 8


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
    assert lib.pn_rk_adjoint_step(None, _lib.PN_F64, 4, ts, None, 0.0, 0.1, None, None, None, vcb, None, None) != 0
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
    rc = lib.pn_rk_adjoint_step(None, _lib.PN_F64, 4, ts, ctypes.byref(ops), 0.0, 0.1, p, p, None, vcb, None, None)
    assert rc != 0 and b"VJP callback failed" in lib.pn_last_error()
    # and a successful walk: rk4 = 3 stage launches + the closing combination, 4 evaluations at t + c_i h
    seen = []
    ok = _lib.STAGE_CB(lambda user, stage, t: seen.append((stage, t)) or p.value)
    _lib.check(lib.pn_rk_attempt(None, _lib.PN_F64, 4, ts, ctypes.byref(ops), 1.0, 0.5, p, p, ys, None, 0, 0.0, ok, None, 0, None, None, kout))
    assert seen == [(0, 1.0), (1, 1.25), (2, 1.25), (3, 1.5)] and len(calls) == 4
    lib.pn_ts_destroy(ts)
