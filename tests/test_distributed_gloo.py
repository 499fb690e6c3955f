"""world_size-2 gloo test of the batch-sharded path: each rank integrates its own
contiguous batch shard; ONE all-reduce of the flat parameter gradient per backward; with an
adaptive method one scalar all-reduce per step attempt keeps the reference's global WRMS norm
(and hence identical step sequences) on every rank.  The device ops are the CPU stand-in
(tests/_cpu_vecops.py); on the GPU box the same code path runs over RCCL (backend "nccl")."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, method, opts, step_size, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from _cpu_vecops import CpuVecOps
    from pnode_amd import options, petsc_adjoint
    from problems import SpiralFunc, SpiralTruth, flat_grads

    options.clear()
    for k, v in opts.items():
        options.set_option(k, v)
    torch.manual_seed(0)
    B = 12
    y0_full = torch.randn(B, 2, dtype=torch.float64)
    t = torch.tensor([0.0, 0.3, 1.0], dtype=torch.float64)
    target_full = torch.randn(3, B, 2, dtype=torch.float64)
    lo, hi = rank * B // world, (rank + 1) * B // world
    f = SpiralFunc() if method == "rk4" else SpiralTruth()      # the truth ODE makes the controller reject
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0_full[lo:hi], f, step_size=step_size, method=method)
    ode.setProcessGroup(None, average=True, global_error_norm=True)
    y = y0_full[lo:hi].clone().requires_grad_(True)
    pred = ode.odeint_adjoint(y, t)
    loss = torch.mean(torch.abs(pred - target_full[:, lo:hi]))
    loss.backward()
    hs = [ode._step_info(k)[1] for k in range(ode._nsteps)]
    torch.save({"pred": pred.detach(), "gy": y.grad, "gtheta": flat_grads(f), "h": hs,
                "rej": ode._lib.pn_ts_rejections(ode._ts)}, out_path % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("method,opts,step_size", [
    ("rk4", {"ts_adapt_type": "none"}, 0.05),
    ("dopri5", {}, 0.1),
    ("bosh3", {"ts_trajectory_max_cps_ram": 3}, 0.1),
])
def test_two_rank_sharding_equals_the_full_batch_solve(tmp_path, method, opts, step_size):
    world = 2
    out = str(tmp_path / "rank%d.pt")
    mp.spawn(_worker, args=(world, _free_port(), method, opts, step_size, out), nprocs=world, join=True)
    parts = [torch.load(out % r) for r in range(world)]

    from oracle.ts_oracle import ODEPetscOracle
    from problems import SpiralFunc, SpiralTruth, flat_grads, rel_err
    torch.manual_seed(0)
    B = 12
    y0 = torch.randn(B, 2, dtype=torch.float64)
    t = torch.tensor([0.0, 0.3, 1.0], dtype=torch.float64)
    target = torch.randn(3, B, 2, dtype=torch.float64)
    f = SpiralFunc() if method == "rk4" else SpiralTruth()
    ref = ODEPetscOracle(dict(opts, oracle_exact_rollback=1))
    ref.setupTS(y0, f, step_size=step_size, method=method)
    y = y0.clone().requires_grad_(True)
    pred = ref.odeint_adjoint(y, t)
    torch.mean(torch.abs(pred - target)).backward()
    te, h, rej = ref.step_log()

    # identical step sequence on every rank = the full-batch (global-norm) sequence
    for p in parts:
        assert len(p["h"]) == len(h) and p["rej"] == rej
        assert torch.allclose(torch.tensor(p["h"], dtype=torch.float64), torch.tensor(h, dtype=torch.float64), rtol=1e-10)
    if method != "rk4":
        assert rej > 0
    # forward: shards concatenate to the full solution
    assert rel_err(torch.cat([p["pred"] for p in parts], dim=1), pred) < 1e-11
    # dL/dtheta: identical on both ranks after the all-reduce, equal to the full-batch gradient
    assert torch.equal(parts[0]["gtheta"], parts[1]["gtheta"])
    assert rel_err(parts[0]["gtheta"], flat_grads(f)) < 1e-10
    # dL/dy0 stays sharded; local mean -> global mean is a factor 1/world
    assert rel_err(torch.cat([p["gy"] for p in parts], dim=0) / world, y.grad) < 1e-10


def _imex_worker(rank, world, port, name, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from _cpu_vecops import CpuVecOps
    from pnode_amd import options, petsc_adjoint
    from problems import DiffusionIM, ReactionEX

    options.clear()
    for k, v in {"ts_adapt_type": "none", "ts_arkimex_type": name, "snes_type": "ksponly"}.items():
        options.set_option(k, v)
    torch.manual_seed(0)
    B, n = 8, 6
    y0_full = torch.randn(B, n, dtype=torch.float64)
    t = torch.tensor([0.0, 0.1, 0.25], dtype=torch.float64)
    target_full = torch.randn(3, B, n, dtype=torch.float64)
    lo, hi = rank * B // world, (rank + 1) * B // world
    fI, fE = DiffusionIM(n), ReactionEX(n)
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0_full[lo:hi], fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE,
                batch_size=hi - lo, linear_solver="torch", matrixfree_jacobian=False)
    ode.setProcessGroup(None, average=True)
    y = y0_full[lo:hi].clone().requires_grad_(True)
    pred = ode.odeint_adjoint(y, t)
    torch.mean(torch.abs(pred - target_full[:, lo:hi])).backward()
    g = torch.cat([p.grad.reshape(-1) for p in list(fI.parameters()) + list(fE.parameters())])
    torch.save({"pred": pred.detach(), "gy": y.grad, "gtheta": g}, out_path % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["3", "l2"])
def test_two_rank_sharding_of_the_imex_direct_solve_path(tmp_path, name):
    """BASELINE config 5's distributed form: IMEX + -snes_type ksponly + linear_solver="torch" has no norm in
    its stage solves, so batch shards are exactly independent: the shards concatenate to the full-batch
    solution of the oracle, dL/dtheta ([IM, EX] order) is the all-reduced sum on both ranks."""
    world = 2
    out = str(tmp_path / "rank%d.pt")
    mp.spawn(_imex_worker, args=(world, _free_port(), name, out), nprocs=world, join=True)
    parts = [torch.load(out % r) for r in range(world)]
    from oracle.arkimex_oracle import odeint_adjoint_arkimex
    from problems import DiffusionIM, ReactionEX, rel_err
    torch.manual_seed(0)
    B, n = 8, 6
    y0 = torch.randn(B, n, dtype=torch.float64)
    t = torch.tensor([0.0, 0.1, 0.25], dtype=torch.float64)
    target = torch.randn(3, B, n, dtype=torch.float64)
    fI, fE = DiffusionIM(n), ReactionEX(n)
    y = y0.clone().requires_grad_(True)
    pred = odeint_adjoint_arkimex(fI, fE, y, t, 0.05, name)
    torch.mean(torch.abs(pred - target)).backward()
    g = torch.cat([p.grad.reshape(-1) for p in list(fI.parameters()) + list(fE.parameters())])
    assert rel_err(torch.cat([p["pred"] for p in parts], dim=1), pred) < 1e-11
    assert torch.equal(parts[0]["gtheta"], parts[1]["gtheta"]) and rel_err(parts[0]["gtheta"], g) < 1e-9
    assert rel_err(torch.cat([p["gy"] for p in parts], dim=0) / world, y.grad) < 1e-9


def _theta_worker(rank, world, port, method, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    res = _theta_solve(method, rank, world)
    torch.save(res, out_path % rank)
    dist.barrier()
    dist.destroy_process_group()


def _theta_solve(method, rank, world):
    from _cpu_vecops import CpuVecOps
    from pnode_amd import options, petsc_adjoint
    from problems import TimeDependent, flat_grads
    options.clear()
    options.set_option("ts_adapt_type", "none")          # Newton / GMRES at PETSc's default (loose) tolerances
    torch.manual_seed(0)
    B, d = 10, 4
    y0_full = torch.randn(B, d, dtype=torch.float64)
    t = torch.tensor([0.0, 0.2, 0.5], dtype=torch.float64)
    target_full = torch.randn(3, B, d, dtype=torch.float64)
    lo, hi = (rank * B // world, (rank + 1) * B // world) if world > 1 else (0, B)
    f = TimeDependent(d)
    ode = petsc_adjoint.ODEPetsc(backend=CpuVecOps)
    ode.setupTS(y0_full[lo:hi], f, step_size=0.1, method=method, implicit_form=True)
    if world > 1:
        ode.setProcessGroup(None, average=False)
    y = y0_full[lo:hi].clone().requires_grad_(True)
    pred = ode.odeint_adjoint(y, t)
    (torch.abs(pred - target_full[:, lo:hi]).sum() / (3 * B * d)).backward()
    return {"pred": pred.detach(), "gy": y.grad, "gtheta": flat_grads(f),
            "its": (ode._theta.newton_its, ode._theta.linear_its)}


@pytest.mark.parametrize("method", ["cn", "beuler"])
def test_two_rank_newton_gmres_follows_the_unsharded_solve(tmp_path, method):
    """Implicit stage solves by Newton-GMRES: norms and Gram-Schmidt products are summed over the ranks, so
    every rank builds the Krylov space of the UNSHARDED system and stops where the single-process solve
    stops -- same Newton and GMRES iteration counts, same numbers to round-off, even at PETSc's loose
    default tolerances (where per-shard convergence would differ visibly)."""
    world = 2
    out = str(tmp_path / "rank%d.pt")
    mp.spawn(_theta_worker, args=(world, _free_port(), method, out), nprocs=world, join=True)
    parts = [torch.load(out % r) for r in range(world)]
    sys.path.insert(0, HERE)
    full = _theta_solve(method, 0, 1)
    from problems import rel_err
    assert parts[0]["its"] == parts[1]["its"] == full["its"]
    assert rel_err(torch.cat([p["pred"] for p in parts], dim=1), full["pred"]) < 1e-12
    assert rel_err(torch.cat([p["gy"] for p in parts], dim=0), full["gy"]) < 1e-10
    assert torch.equal(parts[0]["gtheta"], parts[1]["gtheta"]) and rel_err(parts[0]["gtheta"], full["gtheta"]) < 1e-10
