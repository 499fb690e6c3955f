#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

Workload (the configuration BASELINE.json's metric/north_star is
quoted on): batch 4096 x state_dim 512 fp32, func = 3x[Linear(512,512)+Tanh]+Linear(512,512)
(W ~ N(0,0.02), b = 0), rk4 fixed step h = 0.01, 100 time steps, adjoint on, stages stored.
One bench "step" = one forward sweep + one reverse (discrete adjoint) sweep over the 100 time
steps; value = time-steps/s (fwd+adjoint) summed over ranks (each rank integrates its own
batch shard of 4096 trajectories: weak scaling, one all-reduce of the parameter gradient per
backward over RCCL).  The timed region replays the two sweeps from hipGraphs (--mode eager for
plain stream launches); reference semantics are kept (per stage VJP: one forward + one backward
of func, as pa.py:66-74).

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` covers the solver kernels (the pn_* entry points:
stage AXPYs, adjoint cotangents, lambda update), timed live with HIP events bound to each
dispatch (hipExtLaunchKernelGGL start/stop events); func's GEMMs are PyTorch/hipBLASLt and
are not part of it.  `cpu_baseline` is the oracle (restated PETSc path) on the host cores.
"""
import argparse
import ctypes
import json
import os

# see pnode_amd/__init__.py: hipGraph replays of PyTorch reductions need this on ROCm 7.2
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
ALG_VECTORS_PER_STEP = 32       # rk4 fwd (15) + adjoint (17) vector moves per step


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--nt", type=int, default=100, help="time steps per solve")
    ap.add_argument("--dt", type=float, default=0.01)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra (non-headline) measurements")
    ap.add_argument("--mode", choices=["graph", "eager"], default="graph",
                    help="launch mode of the timed region: hipGraph replay (default) or plain stream launches")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--tunableop", action="store_true",
                    help="let PyTorch's TunableOp pick func's GEMM kernels (tuned in the untimed setup solves; +2-3 %% "
                         "at C3a, profiles/README.md); off by default: the headline uses PyTorch's stock heuristics")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --batch is the GLOBAL batch, split evenly over the ranks (default: weak, "
                         "--batch trajectories per GPU)")
    return ap.parse_args()


def cpu_baseline(args, budget_s):
    """Oracle (C restatement of the PETSc op sequence + per-stage Python callbacks) on the
    host: same state size, same func, same scheme, fewer time steps (bounded sample)."""
    import torch
    from oracle.ts_oracle import ODEPetscOracle
    from problems import MLPFunc

    torch.manual_seed(0)
    f = MLPFunc(args.dim, torch.float32)
    y0 = torch.randn(args.batch, args.dim)

    def solve(nt):
        ode = ODEPetscOracle({"ts_adapt_type": "none", "ts_trajectory_solution_only": 0})
        ode.setupTS(y0, f, step_size=args.dt, method="rk4")
        f.zero_grad()
        y = y0.clone().requires_grad_(True)
        t0 = time.perf_counter()
        out = ode.odeint_adjoint(y, torch.tensor([args.dt * nt]))
        out.abs().mean().backward()
        return time.perf_counter() - t0

    solve(1)                                   # warm-up (allocations, thread pools)
    # func's GEMMs run on torch's intra-op pool: calibrate the pool size (more threads than
    # the 4096x512 GEMMs can feed only adds synchronisation cost), then keep the fastest
    ncpu = os.cpu_count() or 1
    best = None
    for th in sorted({min(ncpu, c) for c in (8, 16, 32, 64, ncpu)}):
        torch.set_num_threads(th)
        solve(1)
        dt1 = solve(1)
        if best is None or dt1 < best[1]:
            best = (th, dt1)
    threads = best[0]
    torch.set_num_threads(threads)
    per = solve(2) / 2.0
    nt = int(max(2, min(args.nt, budget_s / max(per, 1e-6))))
    dt = solve(nt)
    return {"value": nt / dt, "unit": "time-steps/s", "cores": threads, "kind": "port",
            "sample": "%d of %d rk4 time steps fwd+adjoint at batch %d x %d fp32, stages stored; "
                      "vector ops single-threaded C (VecSeq-like), func on %d torch threads "
                      "(fastest of 8..%d on this host)"
                      % (nt, args.nt, args.batch, args.dim, threads, ncpu)}


def pmc_traffic_per_launch():
    """HBM bytes per solver-kernel launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, separate runs, gfx950 FETCH correction applied), summarised in profiles/ by the
    round that measured them; None if no such summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        return json.load(open(files[-1]))["rk4_time_step"]["hbm_bytes_per_launch_avg"]
    except Exception:
        return None


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # PN_BENCH_BACKEND=gloo is a test hook: it lets the multi-rank flow be exercised on a box
    # with fewer GPUs than ranks (ranks then share devices); the real runs use RCCL ("nccl")
    backend = os.environ.get("PN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # the library ships prebuilt in-tree; only a missing one is built, by rank 0
    if not os.path.exists(ge.LIB):
        if rank == 0:
            ge.build_library(force=True)
        else:
            while not os.path.exists(ge.LIB):
                time.sleep(0.5)
            time.sleep(2.0)
    from pnode_amd import _lib, options, petsc_adjoint
    from problems import MLPFunc

    lib = _lib.load()
    base_opts = {"ts_adapt_type": "none", "ts_trajectory_type": "memory", "ts_trajectory_solution_only": "0"}

    if args.tunableop:
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.set_filename(os.path.join("/tmp", "pnode_amd_tunableop_rank%d.csv" % rank))   # results file: scratch
    if args.strong:
        if args.batch % world:
            raise SystemExit("--strong: --batch must be divisible by the number of ranks")
        args.batch //= world
    torch.manual_seed(0)                     # same parameters on every rank
    func = MLPFunc(args.dim, torch.float32).to(dev)
    torch.manual_seed(1234 + rank)           # a different batch shard per rank
    y0 = torch.randn(args.batch, args.dim, device=dev)
    t = torch.tensor([args.dt * args.nt])

    def make_ode(extra):
        options.clear()
        for k, v in dict(base_opts, **extra).items():
            options.set_option(k, v)
        o = petsc_adjoint.ODEPetsc()
        o.setupTS(y0, func, step_size=args.dt, method="rk4", enable_adjoint=True)
        options.clear()
        return o

    def one_solve(o):
        for p in func.parameters():
            p.grad = None
        y = y0.detach().requires_grad_(True)
        out = o.odeint_adjoint(y, t)
        loss = out.abs().mean()
        loss.backward()
        return loss

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(o, k):
        sync()
        t0 = time.perf_counter()
        for _ in range(k):
            one_solve(o)
        sync()
        tv = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tv, op=dist.ReduceOp.MAX)
        return tv.item()

    # ---- headline solver.  mode "graph": the whole forward sweep and the whole reverse sweep
    # are replayed from two hipGraphs (same kernels, same order, bit-identical results; two
    # eager calls + one capturing call happen here, untimed).  Captured BEFORE the process
    # group exists, so no RCCL helper thread is alive while a capture is in progress.
    mode = args.mode
    ode = None
    if mode == "graph":
        try:
            ode = make_ode({"pn_graph_capture": "1"})
            for _ in range(petsc_adjoint.ODEPetsc.GRAPH_WARMUP_CALLS + 1):
                one_solve(ode)
            torch.cuda.synchronize()
            assert ode.graphs_captured
        except Exception as exc:                      # fall back to eager launches, say so
            sys.stderr.write("bench: hipGraph capture failed (%r); falling back to eager launches\n" % (exc,))
            mode = "eager(graph-capture-failed)"
            ode = None
    if ode is None:
        ode = make_ode({})
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        ode.setProcessGroup(None, average=True)
        dist.barrier()

    for _ in range(args.warmup):
        one_solve(ode)
    elapsed = timed(ode, args.steps)
    nsteps = ode.num_steps
    assert nsteps == args.nt, (nsteps, args.nt)

    # ---- roofline pass: the same solve with eager launches, every solver-kernel dispatch
    # bracketed by HIP start/stop events (events cannot be attached to graph nodes).  Same
    # kernels on the same data as the timed region above.
    ode_e = ode if mode.startswith("eager") else make_ode({})
    if world > 1 and ode_e is not ode:
        ode_e.setProcessGroup(None, average=True)
    kr = max(1, min(args.steps, 5))
    one_solve(ode_e)
    sync()
    lib.pn_prof_enable(1)
    elapsed_e = timed(ode_e, kr)
    L = (ctypes.c_int64 * len(_lib.KERNEL_IDS))()
    us = (ctypes.c_double * len(_lib.KERNEL_IDS))()
    by = (ctypes.c_double * len(_lib.KERNEL_IDS))()
    _lib.check(lib.pn_prof_collect(L, us, by))
    lib.pn_prof_enable(0)

    # ---- the one collective of the path, timed alone (all-reduce time)
    allreduce_us = None
    if world > 1:
        flat = torch.zeros(sum(p.numel() for p in func.parameters()), device=dev)
        for _ in range(3):
            dist.all_reduce(flat)
        sync()
        t0 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(flat)
        torch.cuda.synchronize()
        allreduce_us = 1e6 * (time.perf_counter() - t0) / 20

    # ---- extra, NOT the headline (single GPU only)
    variants = None
    if world == 1 and not args.no_variants:
        variants = {"eager": {"value": args.nt * kr / elapsed_e, "unit": "time-steps/s",
                              "note": "plain stream launches, events on (the roofline pass)"}}
        for name, extra, note in [
            ("retain_graph", {"pn_trajectory_retain_graph": "1"},
             "eager launches; stage autograd tapes kept in HBM, no forward recompute of func in the reverse sweep "
             "(the reference recomputes, pa.py:66-74)"),
            ("hip_graph+retain_graph", {"pn_graph_capture": "1", "pn_trajectory_retain_graph": "1"},
             "graph replay + retained tapes"),
        ]:
            try:
                ov = make_ode(extra)
                for _ in range(3):
                    one_solve(ov)
                tv = timed(ov, args.steps)
                variants[name] = {"value": args.nt * args.steps / tv, "unit": "time-steps/s", "note": note}
            except Exception as exc:
                variants[name] = {"error": repr(exc)}
            ov = None
            torch.cuda.empty_cache()

    if rank == 0:
        n = args.batch * args.dim
        w = 4
        solver = (0, 2, 3)                    # pn_rk_stage, pn_adj_theta, pn_adj_accum
        k_usec = sum(us[i] for i in solver)
        k_launch = sum(L[i] for i in solver)
        alg_bytes = float(ALG_VECTORS_PER_STEP) * n * w * args.nt * kr
        achieved = alg_bytes / (k_usec * 1e-6) / 1e9 if k_usec > 0 else 0.0
        # SURVEY 8(d) inclusive accounting: the engine (not autograd) accumulates mu, so add the
        # parameter-sensitivity kernel: 4 stages x (read g, read mu, write mu) x np x w per time step
        n_par = sum(p.numel() for p in func.parameters())
        all_usec = k_usec + us[4]
        all_bytes = alg_bytes + 4.0 * 3.0 * n_par * w * args.nt * kr
        all_achieved = all_bytes / (all_usec * 1e-6) / 1e9 if all_usec > 0 else 0.0
        per_kernel = {}
        for i, name in enumerate(_lib.KERNEL_IDS):
            if L[i]:
                per_kernel[name] = {"launches": int(L[i]), "avg_us": us[i] / L[i],
                                    "GBps_moved": by[i] / (us[i] * 1e-6) / 1e9}
        out = {
            "metric": "time-steps/sec (fwd+adjoint)",
            "value": world * args.nt * args.steps / elapsed,
            "unit": "time-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C3a: MLP dynamics 3x512 tanh, batch %d x state_dim %d per GPU, rk4 fixed h=%g, "
                                   "%d time steps, adjoint on, stages stored in HBM" % (args.batch, args.dim, args.dt, args.nt),
                       "batch_per_gpu": args.batch, "state_dim": args.dim, "time_steps": args.nt,
                       "launch_mode": mode, "tunableop": bool(args.tunableop),
                       "parallelism": "batch-sharded x%d, one RCCL all-reduce of dL/dtheta per backward" % world,
                       "allreduce_us": allreduce_us},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_per_launch(),
                         "kernel": "pn_lincomb_kernel (pn_rk_stage + pn_adj_theta + pn_adj_accum)",
                         "measured_in": "separate eager pass of the same solve (%d solves), HIP start/stop events bound "
                                        "to each dispatch; the timed region above replays hipGraphs" % kr
                                        if not mode.startswith("eager") else "the timed region (eager launches)",
                         "algorithmic_bytes_per_time_step": ALG_VECTORS_PER_STEP * n * w,
                         "solver_kernel_us_per_time_step": k_usec / (args.nt * kr),
                         "avg_launch_us": k_usec / max(k_launch, 1), "launches_per_time_step": k_launch / (args.nt * kr),
                         "with_param_accum": {"achieved": all_achieved, "frac": all_achieved / HBM_PEAK_GBS,
                                              "us_per_time_step": all_usec / (args.nt * kr),
                                              "algorithmic_bytes_per_time_step": ALG_VECTORS_PER_STEP * n * w + 12 * n_par * w,
                                              "note": "all pn_* kernels of the sweep incl. pn_param_accum (SURVEY 8d: + s*3*np*w)"},
                         "per_kernel": per_kernel},
            "variants": variants,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
