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


DRIVERS = "/root/reference/examples-pnode"


def _run_driver(tmp_path, name, args, timeout=900):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg")
    r = subprocess.run([sys.executable, os.path.join(HERE, "ref_harness", "run_driver.py"), os.path.join(DRIVERS, name)] + args,
                       capture_output=True, text=True, timeout=timeout, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    return r.stdout


@pytest.mark.skipif(not os.path.exists(DRIVERS), reason="the reference is only mounted in the build container")
def test_the_references_spiral_demo_runs_unmodified(tmp_path):
    """examples-pnode/ode_demo_petsc.py as its header says to run it (rk4, -ts_adapt_type none, memory trajectory,
    stages kept), a few iterations: truth by ODEPetsc.odeint (1000 steps), training by odeint_adjoint."""
    out = _run_driver(tmp_path, "ode_demo_petsc.py",
                      ["--double_prec", "--niters", "6", "--test_freq", "3", "--method", "rk4", "-ts_adapt_type", "none",
                       "-ts_type", "rk", "-ts_rk_type", "4", "-ts_trajectory_type", "memory", "-ts_trajectory_solution_only", "0"])
    losses = [float(l.split("Total Loss")[1]) for l in out.splitlines() if "Total Loss" in l]
    assert len(losses) == 2 and all(0.0 < x < 5.0 for x in losses), out[-800:]


@pytest.mark.skipif(not os.path.exists(DRIVERS), reason="the reference is only mounted in the build container")
def test_the_references_pendulum_dae_driver_runs_unmodified(tmp_path):
    """examples-pnode/pendulum_DAE.py (Crank-Nicolson, implicit_form, SINGULAR mass matrix, Newton-GMRES) as its
    header says to run it, a few iterations."""
    out = _run_driver(tmp_path, "pendulum_DAE.py",
                      ["--double_prec", "--implicit_form", "--niters", "4", "--test_freq", "2", "-ts_trajectory_type", "memory"])
    lines = [l for l in out.splitlines() if l.startswith("PNODE: Iter")]
    assert len(lines) == 2 and all("NFE-F" in l and "NFE-B" in l for l in lines), out[-800:]


BURGERS = "/root/reference/examples-sinode/Burgers/Burgers.py"


@pytest.mark.skipif(not os.path.exists(BURGERS), reason="the reference is only mounted in the build container")
@pytest.mark.parametrize("variant", ["imex-3", "imex-l2", "cn"])
def test_the_references_burgers_driver_runs_unmodified(tmp_path, variant):
    """examples-sinode/Burgers/Burgers.py with the option sets of its run script (run_a100_512.sh:20-27: ARKIMEX
    type + -snes_type ksponly + --linear_solver torch; cn with the matrix-free solver), one epoch.  The driver loads
    ./Data_T5_IC100_NX1024.p = [u (IC, 51, N), t]; the file is not part of the reference's repository, so a small
    synthetic one of the same layout (10 x 51 x 64 travelling, decaying waves) is written into the working directory."""
    import pickle

    import numpy as np
    IC, T, N = 10, 51, 64
    x = np.linspace(0, 1, N, endpoint=False)
    rng = np.random.default_rng(0)
    u = np.zeros((IC, T, N))
    for i in range(IC):
        a, ph = rng.uniform(0.5, 1.5), rng.uniform(0, 1)
        for k in range(T):
            u[i, k] = a * np.exp(-0.02 * k) * np.sin(2 * np.pi * (x + ph - 0.01 * k))
    pickle.dump([u, np.arange(T) * 0.1], open(tmp_path / "Data_T5_IC100_NX1024.p", "wb"))
    common = ["--adjoint", "--pnode", "--double_prec", "--use_dlpack", "-ts_trajectory_type", "memory", "-ts_adapt_type", "none",
              "--epoch", "1", "--batch_size", "4", "--batch_time", "1", "--test_freq", "1"]
    if variant.startswith("imex"):
        args = common + ["--imex", "-ts_arkimex_type", variant.split("-")[1], "-snes_type", "ksponly", "--linear_solver", "torch"]
    else:
        args = common + ["--method", "cn", "--implicit_form", "--linear_solver", "petsc"]
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg")
    r = subprocess.run([sys.executable, os.path.join(HERE, "ref_harness", "run_driver.py"), BURGERS] + args,
                       capture_output=True, text=True, timeout=1200, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    line = [l for l in r.stdout.splitlines() if l.startswith("Iter ")][-1]
    train, test = float(line.split("Training Loss")[1].split("|")[0]), float(line.split("Testing Loss")[1].split("|")[0])
    assert 0.0 < train < 0.02 and 0.0 < test < 0.02, line


@pytest.mark.skipif(not os.path.exists(DRIVERS), reason="the reference is only mounted in the build container")
def test_the_references_rober_driver_runs_unmodified(tmp_path):
    """examples-pnode/ROBER.py (stiff kinetics, the source of the reference's tests; Crank-Nicolson, implicit_form, min-max
    normalisation) as its header says to run it, a few iterations.  It imports tensorboardX, which this image lacks: the
    harness supplies a writer that drops what it is given."""
    out = _run_driver(tmp_path, "ROBER.py", ["--double_prec", "--implicit_form", "--normalize", "minmax", "--niters", "4",
                                              "--test_freq", "2", "-ts_trajectory_type", "memory"], timeout=1500)
    lines = [l for l in out.splitlines() if l.startswith("PNODE: Iter")]
    assert len(lines) == 2 and all("NFE-F" in l and "NFE-B" in l for l in lines), out[-800:]
    losses = [float(l.split("Total Loss")[1].split("|")[0]) for l in lines]
    assert all(0.0 < x < 1.0 for x in losses) and losses[1] <= losses[0] * 1.05, lines


@pytest.mark.skipif(not os.path.exists(DRIVERS), reason="the reference is only mounted in the build container")
def test_the_references_unstable_spiral_driver_runs_unmodified(tmp_path):
    """examples-pnode/spiral_unstable.py with the command of its header: it trains with Crank-Nicolson and, at every test
    point, compares the discrete-adjoint gradient with the gradient of a second solver by the dot product of the
    normalised gradients (spiral_unstable.py:349-365) -- the reference's own run-time gradient check."""
    out = _run_driver(tmp_path, "spiral_unstable.py",
                      ["-ts_adapt_type", "none", "-ts_trajectory_type", "memory", "--double_prec", "--ref_method", "rk2",
                       "--pnode_method", "cn", "--niters", "4", "--test_freq", "2", "--implicit_form"], timeout=1500)
    dots = [float(l.split("gradients:")[1].split("|")[0]) for l in out.splitlines() if "Dot product of normalized gradients" in l]
    assert dots and all(d > 0.99 for d in dots), out[-800:]


@pytest.mark.skipif(not os.path.exists("/root/reference/ffjord-pnode/lib/layers/cnf.py"), reason="the reference is only mounted in the build container")
def test_the_references_ffjord_cnf_layer_runs_unmodified_and_matches_unrolled_autograd(tmp_path):
    """The vendored FFJORD continuous normalising flow, the reference's third caller family (ffjord-pnode/lib/layers/cnf.py:73-92):
    a flattened tuple state (z, log p), a new func object on every forward, a func that differentiates inside its own
    forward (exact divergence), solver name "dopri5_fixed" that is not in the method map (falls through to PETSc's default
    3bs).  Imported from where it lies, run through the package, compared with autograd through the unrolled
    steps of the same flattened func: fp32 round-off."""
    import json
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "ref_harness", "ffjord_cnf_check.py")], capture_output=True, text=True,
                       timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    for method in ("rk4", "dopri5_fixed"):
        d = res[method]
        assert d["steps"] == 10 and d["grad_norm"] > 0.1
        assert d["z"] < 1e-5 and d["dlogp"] < 1e-5 and d["grad"] < 1e-5, (method, d)


@pytest.mark.skipif(not os.path.exists(DRIVERS), reason="the reference is only mounted in the build container")
@pytest.mark.parametrize("method,nt", [("euler", 1), ("rk4", 2)])
def test_the_references_cifar10_driver_runs_unmodified(tmp_path, method, nt):
    """examples-pnode/train-Cifar10.py -- BASELINE config 4's driver: SqueezeNext-23 with four ODE blocks
    (models/sqnxt_PETSc.py), each with its own ODEPetsc, setupTS before EVERY forward (train-Cifar10.py:121-139), t = [1.0],
    train-mode BatchNorm inside func, enable_adjoint=False at test time -- with the command of its header, one epoch.  The
    image has neither torchvision nor a network: tests/ref_harness/stubs supplies a small synthetic dataset of the CIFAR-10
    shape and no-op torchsummary / tensorboardX.  The driver itself is run from where it lies, unmodified."""
    import math
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg", PN_FAKE_CIFAR_N="32")
    r = subprocess.run([sys.executable, os.path.join(HERE, "ref_harness", "run_driver.py"), os.path.join(DRIVERS, "train-Cifar10.py"),
                        "-ts_adapt_type", "none", "-ts_trajectory_type", "memory", "--num_epochs", "1", "--method", method,
                        "--Nt", str(nt), "--batch_size", "16", "--test_batch_size", "16"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    out = r.stdout.replace("\\r", "\\n")
    train = [float(s.split("Loss:")[1].split()[0]) for s in out.split("Training Epoch [")[1:]]
    test = [float(s.split("Loss:")[1].split()[0]) for s in out.split("Testing Epoch [")[1:]]
    assert len(train) == 2 and len(test) == 2 and all(math.isfinite(x) and 0.0 < x < 1.0 for x in train + test), out[-600:]
    assert "Epoch #1 Cost" in out
