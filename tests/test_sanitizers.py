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
