"""-m gpu: the C ABI stands alone.  A C++ program with no PyTorch and no Python in the process
(tests/native/abi_gpu_client.cpp) links libpnode_amd.so, owns its device buffers (hipMalloc), and runs one
rk4 step and its discrete adjoint, the fused embedded-error kernel and the host-side stepper through
`include/pnode_amd.h` only."""
import os
import shutil
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_plain_cpp_client_of_the_c_abi(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device in this container")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib_dir = os.path.join(ROOT, "pnode_amd", "lib")
    assert os.path.exists(os.path.join(lib_dir, "libpnode_amd.so")), "build the library first (__graft_entry__.build())"
    exe = str(tmp_path / "abi_gpu_client")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(HERE, "native", "abi_gpu_client.cpp"), "-L" + lib_dir, "-lpnode_amd", "-o", exe],
                   check=True, timeout=600)
    env = dict(os.environ, LD_LIBRARY_PATH=lib_dir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "ABI-CLIENT-OK" in r.stdout, r.stdout + r.stderr
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libtorch" not in ldd and "libpython" not in ldd
