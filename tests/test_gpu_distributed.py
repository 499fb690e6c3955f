"""-m gpu: the batch-sharded path with the real HIP backend.  Two ranks share the single GPU of the
test box and talk over gloo (device tensors); on a multi-GPU node the identical code runs one rank
per GPU over RCCL.  Checked against a single-process full-batch solve on the same GPU."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(method, opts):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    from pnode_amd import options
    from problems import SpiralFunc, SpiralTruth
    options.clear()
    for k, v in opts.items():
        options.set_option(k, v)
    torch.manual_seed(0)
    B = 12
    y0 = torch.randn(B, 2, dtype=torch.float64)
    t = torch.tensor([0.0, 0.3, 1.0], dtype=torch.float64)
    target = torch.randn(3, B, 2, dtype=torch.float64)
    f = SpiralFunc() if method == "rk4" else SpiralTruth()
    return y0, t, target, f


def _solve(y0, t, target, f, method, step_size, group_world):
    from pnode_amd import petsc_adjoint
    from problems import flat_grads
    dev = torch.device("cuda:0")
    f = f.to(dev)
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(y0.to(dev), f, step_size=step_size, method=method)
    if group_world > 1:
        ode.setProcessGroup(None, average=True, global_error_norm=True)
    y = y0.to(dev).requires_grad_(True)
    pred = ode.odeint_adjoint(y, t.to(dev))
    torch.mean(torch.abs(pred - target.to(dev))).backward()
    return {"pred": pred.detach().cpu(), "gy": y.grad.cpu(), "gtheta": flat_grads(f).cpu(),
            "h": [h for _, h in ode.step_log()], "rej": ode.num_rejections}


def _worker(rank, world, port, method, opts, step_size, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    y0, t, target, f = _setup(method, opts)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = rank * y0.shape[0] // world, (rank + 1) * y0.shape[0] // world
    res = _solve(y0[lo:hi], t, target[:, lo:hi], f, method, step_size, world)
    torch.save(res, out_path % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("method,opts,step_size", [
    ("rk4", {"ts_adapt_type": "none"}, 0.05),
    ("dopri5", {}, 0.1),
])
def test_two_ranks_on_the_hip_backend_equal_the_full_batch_solve(tmp_path, method, opts, step_size):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device in this container")
    world = 2
    out = str(tmp_path / "rank%d.pt")
    mp.spawn(_worker, args=(world, _free_port(), method, opts, step_size, out), nprocs=world, join=True)
    parts = [torch.load(out % r) for r in range(world)]
    y0, t, target, f = _setup(method, opts)
    full = _solve(y0, t, target, f, method, step_size, 1)
    from problems import rel_err
    for p in parts:
        assert len(p["h"]) == len(full["h"]) and p["rej"] == full["rej"]
        assert torch.allclose(torch.tensor(p["h"], dtype=torch.float64), torch.tensor(full["h"], dtype=torch.float64), rtol=1e-10)
    if method != "rk4":
        assert full["rej"] > 0
    assert rel_err(torch.cat([p["pred"] for p in parts], dim=1), full["pred"]) < 1e-11
    assert torch.equal(parts[0]["gtheta"], parts[1]["gtheta"])
    assert rel_err(parts[0]["gtheta"], full["gtheta"]) < 1e-10
    assert rel_err(torch.cat([p["gy"] for p in parts], dim=0) / world, full["gy"]) < 1e-10


@pytest.mark.parametrize("strong", [False, True])
def test_bench_contract_with_two_ranks(strong):
    """bench.py launched exactly as the driver launches it for N > 1 (torch.distributed.run, one rank
    per process), with the gloo test hook so that two ranks can share this box's single GPU: one JSON
    line from rank 0 with the whole-job rate, the roofline object and the collective's time."""
    import json
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip("no HIP device in this container")
    env = dict(os.environ, PN_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "256", "--dim", "64", "--nt", "6",
           "--no-cpu-baseline", "--no-variants"] + (["--strong"] if strong else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "time-steps/s"
    assert d["scaling"] == ("strong" if strong else "weak") and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["config"]["batch_per_gpu"] == (128 if strong else 256) and d["config"]["launch_mode"] == "graph"
    assert d["config"]["allreduce_us"] > 0 and d["cpu_baseline"] is None
    assert d["value"] == pytest.approx(2 * 6 * 2 / (d["ms_per_step"] * 2 / 1e3), rel=1e-6)
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["bound"] == "hbm"
