"""-m gpu: the device-resident GMRES (pn_krylov_*, include/pnode_amd.h section 3c) and the replayed linearisations of
the Newton-Krylov stage solves (pnode_amd/theta.py _OpGraph) on the HIP device -- the reference's default
linear_solver="petsc" for TS types BE / CN / ARKIMEX (reference pnode/petsc_adjoint.py:547, 581, 651-656, 701-702)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import require_gpu
from pnode_amd import options, petsc_adjoint
from problems import DiffusionIM, MLPFunc, ReactionEX, TimeDependent, flat_grads, rel_err

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _ops(dtype, n):
    from pnode_amd.petsc_adjoint import HipVecOps
    return HipVecOps(require_gpu(), dtype, n)


def _gmres(ops, kr, apply_op, rhs, x, rtol, chunk, maxit=10000):
    """The host side of a solve, as theta.py drives it; returns (iterations, stop, synchronisations)."""
    ops.lincomb(x, [rhs], [0.0])
    ops.krylov_begin(kr, rhs, rtol, 1e-50, maxit, True)
    k, syncs, m = 0, 0, kr.m
    r = torch.empty_like(rhs)
    while True:
        for _ in range(chunk):
            if k >= m:
                break
            apply_op(kr.vin, kr.w)
            ops.krylov_step(kr, k)
            k += 1
        ops.krylov_close(kr, x)
        stop, kdone, total, res = ops.krylov_status(kr)
        syncs += 1
        if stop:
            return total, stop, syncs
        if kdone >= m:
            ops.copy(kr.vin, x)
            apply_op(kr.vin, kr.w)
            ops.lincomb(r, [rhs, kr.w], [1.0, -1.0])
            ops.krylov_begin(kr, r, rtol, 1e-50, maxit, False)
            k = 0


@pytest.mark.parametrize("dtype,rtol,tol", [(torch.float64, 1e-11, 1e-9), (torch.float32, 1e-5, 2e-4)])
@pytest.mark.parametrize("n", [3, 257, 4099, 64 * 1024, 4096 * 512 + 5])
@pytest.mark.parametrize("restart,chunk", [(30, 1), (30, 7), (4, 3)])
def test_device_gmres_solves_a_nonsymmetric_system(dtype, rtol, tol, n, restart, chunk):
    """A v = d*v + U (W^T v) (diagonal + rank 3, nonsymmetric) through the kernels alone: converges to the dense answer
    (small n) / to a small true residual (large n), with restarts (restart length 4), ragged sizes, and the same
    iteration count whether the host looks after every iteration or after chunks of 7 (launches past convergence are
    no-ops)."""
    ops = _ops(dtype, n)
    dev = ops.device
    g = torch.Generator().manual_seed(n + restart)
    d = (2.0 + torch.rand(n, generator=g, dtype=torch.float64)).to(dev, dtype)
    U = (torch.randn(3, n, generator=g, dtype=torch.float64) / n ** 0.5).to(dev, dtype)
    W = (torch.randn(3, n, generator=g, dtype=torch.float64) / n ** 0.5).to(dev, dtype)
    rhs = torch.randn(n, generator=g, dtype=torch.float64).to(dev, dtype)

    def apply_op(v, w):
        out = d * v[:n] + (W @ v[:n]) @ U
        ops.copy(w, out.contiguous())

    kr = ops.krylov_new(restart)
    x = torch.zeros(n, dtype=dtype, device=dev)
    its, stop, syncs = _gmres(ops, kr, apply_op, rhs, x, rtol, chunk)
    assert stop == 1 and its >= 2
    res = rhs.double() - (d.double() * x.double() + (W.double() @ x.double()) @ U.double())
    assert float(res.norm() / rhs.double().norm()) < tol
    if n <= 4099:
        A = torch.diag(d.double()) + U.double().T @ W.double()
        assert rel_err(x, torch.linalg.solve(A, rhs.double())) < tol * 10
    # same decisions whatever the chunking
    x2 = torch.zeros(n, dtype=dtype, device=dev)
    its2, stop2, syncs2 = _gmres(ops, kr, apply_op, rhs, x2, rtol, 1 if chunk > 1 else 5)
    assert (its2, stop2) == (its, stop) and torch.equal(x, x2)
    if chunk == 7 and restart == 30:
        assert syncs < its or its <= 7


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("n", [64 * 1024, 4096 * 512 + 5])
def test_fused_krylov_products_leave_the_ticket_area_clean(dtype, n):
    """in the form that hipGraphs replay -- pn_krylov_step(k = -1): "the iteration the state block
    says is due", grid sized for the longest column -- the groups a shorter column does not need used to leave without
    drawing a ticket, and one of them dispatched after the deciding block had advanced the column index could join the
    NEXT column's count and leave an arrival counter at one for good.  Now every block of the grid is counted.  Large n
    (the grid exceeds what is resident at once), restart 32, many solves in the k = -1 form with chunks enqueued ahead:
    the ticket area is all-zero after every solve, and iterations and solution are those of the explicit-k form, bit for
    bit."""
    restart = 32
    ops = _ops(dtype, n)
    dev = ops.device
    g = torch.Generator().manual_seed(n)
    d = (2.0 + torch.rand(n, generator=g, dtype=torch.float64)).to(dev, dtype)
    U = (torch.randn(6, n, generator=g, dtype=torch.float64) / n ** 0.5).to(dev, dtype)
    W = (torch.randn(6, n, generator=g, dtype=torch.float64) / n ** 0.5).to(dev, dtype)

    def apply_op(v, w):
        ops.copy(w, (d * v[:n] + (W @ v[:n]) @ U).contiguous())

    def solve(rhs, x, fused, chunk):
        kr = solve.kr
        ops.lincomb(x, [rhs], [0.0])
        ops.krylov_begin(kr, rhs, rtol, 1e-50, 10000, True)
        k = 0
        r = torch.empty_like(rhs)
        while True:
            for _ in range(chunk):
                if k >= restart:
                    break
                apply_op(kr.vin, kr.w)
                ops.krylov_step(kr, -1 if fused else k)
                k += 1
            ops.krylov_close(kr, x)
            stop, kdone, total, res = ops.krylov_status(kr)
            if stop:
                return total, stop
            if kdone >= restart:
                ops.copy(kr.vin, x)
                apply_op(kr.vin, kr.w)
                ops.lincomb(r, [rhs, kr.w], [1.0, -1.0])
                ops.krylov_begin(kr, r, rtol, 1e-50, 10000, False)
                k = 0

    from pnode_amd._lib import load
    rtol = 1e-11 if dtype == torch.float64 else 1e-5
    solve.kr = ops.krylov_new(restart)
    nticket = 33 * 16                                    # (kTicketShards + 1) * kTicketStride doubles (pn_device.h)
    for trial in range(6):
        rhs = torch.randn(n, generator=g, dtype=torch.float64).to(dev, dtype)
        x1 = torch.zeros(n, dtype=dtype, device=dev)
        x2 = torch.zeros(n, dtype=dtype, device=dev)
        its1, stop1 = solve(rhs, x1, False, 3)
        assert not solve.kr.state[:nticket].view(torch.int64).any()
        its2, stop2 = solve(rhs, x2, True, 9)            # k = -1, nine iterations enqueued ahead of every look
        torch.cuda.synchronize()
        assert not solve.kr.state[:nticket].view(torch.int64).any(), "an arrival counter was left non-zero"
        assert (its1, stop1) == (its2, stop2) and stop1 == 1 and torch.equal(x1, x2)


def test_device_gmres_flags_zero_rhs_breakdown_nan_and_the_iteration_limit():
    n = 1000
    ops = _ops(torch.float64, n)
    dev = ops.device
    kr = ops.krylov_new(10)
    x = torch.ones(n, dtype=torch.float64, device=dev)
    ident = lambda v, w: ops.copy(w, v)
    # zero right-hand side: converged at once, x = 0
    its, stop, _ = _gmres(ops, kr, ident, torch.zeros(n, dtype=torch.float64, device=dev), x, 1e-8, 3)
    assert (its, stop) == (0, 1) and float(x.abs().max()) == 0.0
    # A = I: one iteration, happy ending (residual estimate exactly 0 -> converged)
    rhs = torch.randn(n, dtype=torch.float64, device=dev)
    its, stop, _ = _gmres(ops, kr, ident, rhs, x, 1e-8, 3)
    assert its == 1 and stop in (1, 2) and rel_err(x, rhs) < 1e-14
    # A = 0: singular on its Krylov space -> breakdown at the back substitution
    zero = lambda v, w: ops.lincomb(w, [v], [0.0])
    its, stop, _ = _gmres(ops, kr, zero, rhs, x, 1e-8, 2)
    assert stop == 5
    # NaN in the operator
    nan = lambda v, w: ops.lincomb(w, [v], [float("nan")])
    its, stop, _ = _gmres(ops, kr, nan, rhs, x, 1e-8, 2)
    assert stop == 4
    # iteration limit
    g = torch.Generator().manual_seed(0)
    M = torch.randn(n, n, generator=g, dtype=torch.float64).to(dev)
    mat = lambda v, w: ops.copy(w, (M @ v[:n]).contiguous())
    its, stop, _ = _gmres(ops, kr, mat, rhs, x, 1e-14, 4, maxit=23)
    assert (its, stop) == (23, 3)


def _theta_case(method, opts, dtype=torch.float64, B=64, d=24, nt=5, func="time", seed=0, graphs_iters=1):
    dev = require_gpu()
    options.clear()
    for k, v in dict({"ts_adapt_type": "none"}, **opts).items():
        options.set_option(k, v)
    torch.manual_seed(seed)
    y0 = torch.randn(B, d, dtype=torch.float64).to(dev, dtype)
    t = torch.tensor([0.0, 0.1 * (nt // 2), 0.1 * nt], dtype=torch.float64)
    target = torch.randn(3, B, d, dtype=torch.float64).to(dev, dtype)
    ode = petsc_adjoint.ODEPetsc()
    if method == "imex":
        fI, fE = DiffusionIM(d, dtype).to(dev), ReactionEX(d, dtype).to(dev)
        ode.setupTS(y0, fI, step_size=0.1, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=B)
        mods = (fI, fE)
    else:
        f = (TimeDependent(d, dtype) if func == "time" else MLPFunc(d, dtype, std=0.3)).to(dev)
        ode.setupTS(y0, f, step_size=0.1, method=method, implicit_form=True)
        mods = (f,)
    options.clear()
    outs = []
    for it in range(graphs_iters):
        for m in mods:
            m.zero_grad()
        y = y0.clone().requires_grad_(True)
        p = ode.odeint_adjoint(y, t.to(dev))
        torch.mean(torch.abs(p - target)).backward()
        outs.append((p.detach().clone(), y.grad.clone(), torch.cat([flat_grads(m) for m in mods]).clone(),
                     (ode._theta.newton_its, ode._theta.linear_its)))
        with torch.no_grad():                       # a training step: the parameters change IN PLACE
            for m in mods:
                for q in m.parameters():
                    if q.grad is not None:
                        q.add_(q.grad, alpha=-0.05)
    return outs, ode


@pytest.mark.parametrize("method", ["cn", "beuler", "imex"])
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-11), (torch.float32, 2e-4)])
def test_device_gmres_takes_the_decisions_of_the_host_loop_on_the_gpu(method, dtype, tol):
    """Same Newton / GMRES iteration counts and the same numbers (to round-off) as round 2's host-driven loop, eager
    operator in both (graphs off), with fewer host synchronisations than Krylov iterations."""
    tight = {"ksp_rtol": 1e-10, "snes_rtol": 1e-12} if dtype == torch.float64 else {}
    dev_out, dev_ode = _theta_case(method, dict(tight, pn_krylov="device", pn_krylov_graph=0), dtype)
    host_out, host_ode = _theta_case(method, dict(tight, pn_krylov="host", pn_krylov_graph=0), dtype)
    if dtype == torch.float64:
        assert dev_out[0][3] == host_out[0][3]
    else:                                            # fp32 at loose tolerances: a knife-edge decision may flip
        assert abs(dev_out[0][3][1] - host_out[0][3][1]) <= 2 and dev_out[0][3][0] == host_out[0][3][0]
    for a, b in zip(dev_out[0][:3], host_out[0][:3]):
        assert rel_err(a, b) < tol
    th = dev_ode._theta
    assert th.linear_its > 20 and 0 < th.host_syncs < th.linear_its and host_ode._theta.host_syncs == 0


@pytest.mark.parametrize("method,func", [("cn", "time"), ("beuler", "time"), ("cn", "mlp"), ("imex", "time")])
def test_replayed_linearisations_equal_the_eager_operator_over_a_training_loop(method, func):
    """-pn_krylov_graph 1 (the default on the device): the linearisation of f at every stage time and the operator
    products are replayed from hipGraphs captured in the first solve.  Over four solves with the parameters updated in
    place between them, states and gradients equal the eager-operator run (double-VJP in both: same arithmetic) to
    round-off; the dynamics depend on t, so every stage time has its own capture, and later solves capture nothing."""
    opts = {"ksp_rtol": 1e-10, "snes_rtol": 1e-12, "pn_jvp": "double_vjp"}
    g_out, g_ode = _theta_case(method, dict(opts, pn_krylov_graph=1), func=func, graphs_iters=4)
    caps_after = g_ode._theta._op_stats[1]
    e_out, e_ode = _theta_case(method, dict(opts, pn_krylov_graph=0), func=func, graphs_iters=4)
    assert e_ode._theta._op_stats[1] == 0 and caps_after > 0
    for a, b in zip(g_out, e_out):
        assert a[3] == b[3]
        for u, v in zip(a[:3], b[:3]):
            assert rel_err(u, v) < 1e-10
    assert len({tuple(o[3]) for o in g_out}) >= 1 and g_out[0][3][1] > 20
    # later solves replay: the number of captured linearisations stopped growing after the first solve
    st = g_ode._theta
    nt = 5
    expected = (nt if method != "imex" else None)
    if expected is not None:
        assert st._op_stats[1] == 2 * nt                       # one per stage time, forward and transposed
    assert st._op_stats[0] > st._op_stats[1] * 3


@pytest.mark.parametrize("method", ["cn", "beuler"])
def test_forward_mode_product_graphs_equal_the_eager_forward_mode_operator(method):
    """-pn_krylov_graph_form jvp: the product graph is one forward-mode pass of func (torch.func.jvp), the form `auto` picks
    when it is the fastest of the three (eager, double-VJP graph, forward-mode graph).  Same arithmetic as the eager
    operator's default (forward mode): same iteration counts, numbers to round-off, over a training loop."""
    opts = {"ksp_rtol": 1e-10, "snes_rtol": 1e-12}
    g_out, g_ode = _theta_case(method, dict(opts, pn_krylov_graph=1, pn_krylov_graph_form="jvp"), graphs_iters=3)
    e_out, e_ode = _theta_case(method, dict(opts, pn_krylov_graph=0), graphs_iters=3)
    assert g_ode._theta._op_stats[1] > 0 and e_ode._theta._fwd_mode is True
    assert any(e.fwd for e in g_ode._theta._op_graphs.values()) and any(e.transpose for e in g_ode._theta._op_graphs.values())
    for a, b in zip(g_out, e_out):
        assert a[3] == b[3]
        for u, v in zip(a[:3], b[:3]):
            assert rel_err(u, v) < 1e-10


def test_auto_mode_times_the_graphs_once_and_keeps_a_consistent_choice():
    """-pn_krylov_graph auto (the default): at the first capture replay is timed against eager launches (and against the
    forward-mode graph); whatever wins is used for the whole life of the solver, results equal the eager operator's to
    round-off either way."""
    opts = {"ksp_rtol": 1e-10, "snes_rtol": 1e-12}
    a_out, a_ode = _theta_case("cn", dict(opts), graphs_iters=3)
    e_out, e_ode = _theta_case("cn", dict(opts, pn_krylov_graph=0), graphs_iters=3)
    th = a_ode._theta
    assert th._calibrated and th._calibration is not None and th._calibration[0] > 0 and th._calibration[1] > 0
    assert (th._graph_mode == 0) == (th._graphs_dropped is not None)
    for a, b in zip(a_out, e_out):
        assert abs(a[3][1] - b[3][1]) <= 2 and a[3][0] == b[3][0]
        for u, v in zip(a[:3], b[:3]):
            assert rel_err(u, v) < 1e-8


def test_replayed_linearisations_against_the_exact_newton_oracle():
    from oracle.theta_oracle import odeint_adjoint_theta
    dev = require_gpu()
    torch.manual_seed(0)
    y0 = torch.randn(64, 6, dtype=torch.float64)
    t = torch.tensor([0.0, 0.3, 0.5, 1.0], dtype=torch.float64)
    target = torch.randn(4, 64, 6, dtype=torch.float64)
    for k, v in {"ts_adapt_type": "none", "snes_rtol": 1e-13, "snes_stol": 1e-14, "ksp_rtol": 1e-12, "pn_krylov_graph": 1}.items():
        options.set_option(k, v)
    f = TimeDependent(6).to(dev)
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(y0.to(dev), f, step_size=0.1, method="cn", implicit_form=True)
    for _ in range(2):                                # the second solve replays everything
        f.zero_grad()
        y = y0.to(dev).requires_grad_(True)
        p = ode.odeint_adjoint(y, t.to(dev))
        torch.mean(torch.abs(p - target.to(dev))).backward()
    assert ode._theta._op_stats[1] == 20
    rows = slice(0, 4)
    f2 = TimeDependent(6)
    y2 = y0[rows].clone().requires_grad_(True)
    p2 = odeint_adjoint_theta(f2, y2, t, 0.1, "cn")
    assert rel_err(p[:, rows], p2) < 1e-10
    (torch.abs(p2 - target[:, rows]).sum() / target.numel()).backward()
    assert rel_err(y.grad[rows], y2.grad) < 1e-9


def test_a_func_that_cannot_be_captured_falls_back_to_eager_launches():
    """A func that synchronises with the host (.item()) cannot be captured: one warning, then eager launches with the
    same results as -pn_krylov_graph 0."""
    import torch.nn as nn
    dev = require_gpu()

    class Syncing(nn.Module):
        def __init__(self):
            super().__init__()
            self.W = nn.Parameter(torch.eye(5, dtype=torch.float64) * -0.5)

        def forward(self, t, y):
            s = float(y.abs().max().item())          # host synchronisation
            return torch.tanh(y @ self.W) * (1.0 + 0.0 * s)

    def run(graph):
        options.clear()
        for k, v in {"ts_adapt_type": "none", "pn_krylov_graph": graph, "ksp_rtol": 1e-10}.items():
            options.set_option(k, v)
        torch.manual_seed(0)
        y0 = torch.randn(8, 5, dtype=torch.float64, device=dev)
        f = Syncing().to(dev)
        ode = petsc_adjoint.ODEPetsc()
        ode.setupTS(y0, f, step_size=0.1, method="beuler", implicit_form=True)
        y = y0.clone().requires_grad_(True)
        ode.odeint_adjoint(y, torch.tensor([0.3], dtype=torch.float64)).abs().mean().backward()
        return y.grad.clone(), f.W.grad.clone()

    with pytest.warns(RuntimeWarning, match="launch func eagerly"):
        a = run(1)
    b = run(0)
    assert rel_err(a[0], b[0]) < 1e-12 and rel_err(a[1], b[1]) < 1e-12


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_solve(method, lo, hi, world):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    from pnode_amd import options as opt, petsc_adjoint as pa
    from problems import TimeDependent as TD, flat_grads as fg
    dev = torch.device("cuda:0")
    opt.clear()
    opt.set_option("ts_adapt_type", "none")
    torch.manual_seed(0)
    B, d = 12, 6
    y0 = torch.randn(B, d, dtype=torch.float64)
    t = torch.tensor([0.0, 0.2, 0.5], dtype=torch.float64)
    target = torch.randn(3, B, d, dtype=torch.float64)
    f = TD(d).to(dev)
    ode = pa.ODEPetsc()
    ode.setupTS(y0[lo:hi].to(dev), f, step_size=0.1, method=method, implicit_form=True)
    if world > 1:
        ode.setProcessGroup(None, average=False, global_error_norm=True)
    y = y0[lo:hi].to(dev).requires_grad_(True)
    pred = ode.odeint_adjoint(y, t.to(dev))
    (torch.abs(pred - target[:, lo:hi].to(dev)).sum() / target.numel()).backward()
    return {"pred": pred.detach().cpu(), "gy": y.grad.cpu(), "gtheta": fg(f).cpu(),
            "its": (ode._theta.newton_its, ode._theta.linear_its), "syncs": ode._theta.host_syncs}


def _rank_worker(rank, world, port, method, out_path):
    import faulthandler
    faulthandler.dump_traceback_later(240, exit=True)      # a hung collective must not hang the box: dump and leave
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = _rank_solve(method, rank * 12 // world, (rank + 1) * 12 // world, world)
    torch.save(res, out_path % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("method", ["cn", "beuler"])
def test_two_rank_device_gmres_follows_the_unsharded_solve_on_the_gpu(tmp_path, method):
    """Sharded batch, global Krylov space: the Gram-Schmidt products are summed over the ranks on the device (all-reduce of
    the product block between the deferred parts of pn_krylov_step), so both ranks take the decisions of the unsharded
    solve -- same Newton and GMRES iteration counts, same numbers to round-off."""
    require_gpu()
    world = 2
    out = str(tmp_path / "rank%d.pt")
    mp.spawn(_rank_worker, args=(world, _free_port(), method, out), nprocs=world, join=True)
    parts = [torch.load(out % r) for r in range(world)]
    full = _rank_solve(method, 0, 12, 1)
    assert parts[0]["its"] == parts[1]["its"] == full["its"]
    assert rel_err(torch.cat([p["pred"] for p in parts], dim=1), full["pred"]) < 1e-12
    assert rel_err(torch.cat([p["gy"] for p in parts], dim=0), full["gy"]) < 1e-10
    assert torch.equal(parts[0]["gtheta"], parts[1]["gtheta"]) and rel_err(parts[0]["gtheta"], full["gtheta"]) < 1e-10


def test_a_func_changed_behind_the_captures_back_is_noticed():
    """A capture cannot see a change of func that is not a change of a tensor's contents (here: a coefficient kept as a
    Python float).  The first replay of every solve is checked against one eager evaluation: the stale graphs are dropped
    with a warning and the solve is right."""
    import torch.nn as nn
    dev = require_gpu()

    class Scaled(nn.Module):
        def __init__(self):
            super().__init__()
            self.W = nn.Parameter(torch.eye(5, dtype=torch.float64) * -0.5 + 0.1)
            self.gain = 1.0                               # a plain Python attribute

        def forward(self, t, y):
            return self.gain * torch.tanh(y @ self.W)

    def run(graph, change):
        options.clear()
        for k, v in {"ts_adapt_type": "none", "pn_krylov_graph": graph, "ksp_rtol": 1e-10}.items():
            options.set_option(k, v)
        torch.manual_seed(0)
        y0 = torch.randn(8, 5, dtype=torch.float64, device=dev)
        f = Scaled().to(dev)
        ode = petsc_adjoint.ODEPetsc()
        ode.setupTS(y0, f, step_size=0.1, method="beuler", implicit_form=True)
        options.clear()
        outs = []
        for it in range(3):
            if change and it == 2:
                f.gain = 1.7
            y = y0.clone().requires_grad_(True)
            f.zero_grad()
            ode.odeint_adjoint(y, torch.tensor([0.3], dtype=torch.float64)).abs().mean().backward()
            outs.append((y.grad.clone(), f.W.grad.clone()))
        return outs, ode

    ref, _ = run(0, True)
    with pytest.warns(RuntimeWarning, match="no longer computes what was captured"):
        got, ode = run(1, True)
    for a, b in zip(got, ref):
        assert rel_err(a[0], b[0]) < 1e-10 and rel_err(a[1], b[1]) < 1e-10
    assert ode._theta._graph_mode == 0
    same, ode2 = run(1, False)                             # nothing changed: the graphs stay
    assert ode2._theta._graph_mode == 1 and ode2._theta._op_stats[1] > 0
