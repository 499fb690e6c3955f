"""The C++ host engine (stepper state machine, checkpoint scheduler, GMRES core) compiled with
AddressSanitizer + UndefinedBehaviorSanitizer on the CPU and driven by tests/native/host_selftest.cpp
(GPU sanitizers are not available on this pool; the device kernels are covered by parity tests)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_engine_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_selftest")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "pnode_amd", "csrc"),
           os.path.join(ROOT, "tests", "native", "host_selftest.cpp"),
           os.path.join(ROOT, "pnode_amd", "csrc", "pn_ts.cpp"), "-o", exe]
    subprocess.run(cmd, check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([exe], env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host selftest ok" in out.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
@pytest.mark.parametrize("sanitizer", ["address,undefined", "thread"])
def test_disk_tier_engine_under_sanitizers(tmp_path, sanitizer):
    """pn_spill.cpp is the one multithreaded component (caller + I/O thread, mutex + two condition variables): host
    mode (device = 0, no HIP call is executed) under ASan+UBSan and under ThreadSanitizer."""
    hip_inc = "/opt/rocm/include"
    hip_lib = "/opt/rocm/lib"
    if not os.path.exists(os.path.join(hip_inc, "hip", "hip_runtime_api.h")):
        pytest.skip("HIP headers not installed")
    exe = str(tmp_path / "spill_selftest")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + sanitizer, "-fno-sanitize-recover=all", "-D__HIP_PLATFORM_AMD__",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "pnode_amd", "csrc"), "-I" + hip_inc,
           os.path.join(ROOT, "tests", "native", "spill_selftest.cpp"),
           os.path.join(ROOT, "pnode_amd", "csrc", "pn_spill.cpp"), os.path.join(ROOT, "pnode_amd", "csrc", "pn_ts.cpp"),
           "-L" + hip_lib, "-lamdhip64", "-Wl,-rpath," + hip_lib, "-pthread", "-o", exe]
    subprocess.run(cmd, check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1", TSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([exe, str(tmp_path)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "spill selftest ok" in out.stdout
