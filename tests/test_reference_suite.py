"""The reference's OWN test file, unmodified, against pnode_amd.

/root/reference/tests/test_pnode.py holds the reference's integration tests (explicit RK, Crank-Nicolson, IMEX on
ROBER with its known-answer asserts).  It exists only in the build container, so this test is skipped elsewhere;
here pytest is pointed at the file where it lies (nothing is copied), with tests/ref_harness/pnode_amd_ref_plugin.py
supplying the import environment: `petsc4py` -> compat shim, `pnode` -> shim package, device entry points -> the CPU
stand-in (there is no GPU in this container; the GPU parity tests assert the same constants on the HIP path)."""
import os
import subprocess
import sys

import pytest

REF_TEST = "/root/reference/tests/test_pnode.py"
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.exists(REF_TEST), reason="the reference is only mounted in the build container")
def test_the_references_own_tests_pass_unmodified(tmp_path):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1",
               PYTHONPATH=os.pathsep.join([os.path.join(HERE, "ref_harness"), os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, "-m", "pytest", "-p", "pnode_amd_ref_plugin", "-p", "no:cacheprovider", "-q",
                        "--rootdir", str(tmp_path), "-c", os.devnull, REF_TEST],
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    tail = r.stdout[-1500:] + r.stderr[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout and "error" not in r.stdout.lower().replace("errors", ""), tail
    npassed = int(r.stdout.strip().splitlines()[-1].split(" passed")[0].split()[-1])
    assert npassed >= 3, tail
