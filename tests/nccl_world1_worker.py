"""Worker of tests/test_gpu_distributed.py::test_rccl_world_size_one_full_flow (run as a fresh process).

Initialises the REAL multi-GPU backend ("nccl" = RCCL on ROCm) with world_size 1 on the one GPU of the
test box and runs the whole sharded flow on it: setProcessGroup, eager solves, whole-sweep hipGraph
capture while the RCCL communicator and its watchdog thread are alive, graph replays, and the ncclAllReduce
of dL/dtheta after every backward (adaptive case: the per-attempt scalar all-reduce of the error norm; implicit case:
the all-reduce of the Gram-Schmidt products of the device-resident GMRES).
Prints one JSON line with the largest difference to the same solves without a process group."""
import json
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from pnode_amd import options, petsc_adjoint  # noqa: E402
from problems import MLPFunc, flat_grads  # noqa: E402


def solves(group, method, opts, n_calls):
    dev = torch.device("cuda:0")
    options.clear()
    for k, v in opts.items():
        options.set_option(k, v)
    torch.manual_seed(0)
    f = MLPFunc(64, torch.float32).to(dev)
    y0 = torch.randn(512, 64, device=dev)
    t = torch.tensor([0.0, 0.1, 0.3])
    ode = petsc_adjoint.ODEPetsc()
    ode.setupTS(y0, f, step_size=0.02, method=method, implicit_form=method in ("cn", "beuler"))
    options.clear()
    if group:
        ode.setProcessGroup(None, average=True, global_error_norm=True)
    outs = []
    for it in range(n_calls):
        for p in f.parameters():
            p.grad = None
        y = (y0 * (1.0 + 0.05 * it)).requires_grad_(True)
        out = ode.odeint_adjoint(y, t)
        out.abs().mean().backward()
        outs.append(torch.cat([out.detach().reshape(-1), y.grad.reshape(-1), flat_grads(f)]).clone())
    torch.cuda.synchronize()
    return outs, ode


def main():
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1])
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    probe = torch.ones(1024, device="cuda:0")
    dist.all_reduce(probe)                       # creates the communicator, issues a real ncclAllReduce
    torch.cuda.synchronize()
    res = {"backend": dist.get_backend(), "probe": float(probe.sum())}
    cases = {"rk4_graph": ("rk4", {"ts_adapt_type": "none", "ts_trajectory_solution_only": 0, "pn_graph_capture": 1}, 5),
             "dopri5_global_norm": ("dopri5", {}, 2),
             # Newton-GMRES with its state on the device: the Gram-Schmidt products go through ncclAllReduce (on the device,
             # in stream order) between the deferred parts of pn_krylov_step; linearisations replayed from hipGraphs
             # (fp32 states: Newton tolerances an fp32 residual can reach)
             "cn_krylov": ("cn", {"ts_adapt_type": "none", "pn_krylov_graph": 1, "snes_rtol": 1e-5, "snes_stol": 1e-6}, 3)}
    for name, (method, opts, n) in cases.items():
        with_pg, ode = solves(True, method, opts, n)
        res[name + "_graphs"] = bool(ode.graphs_captured)
        res[name + "_world"] = ode._world()
        without, _ = solves(False, method, opts, n)
        res[name] = max(float((a - b).abs().max()) for a, b in zip(with_pg, without))
        if ode._theta is not None:
            res[name + "_its"] = [ode._theta.newton_its, ode._theta.linear_its, ode._theta.host_syncs, ode._theta._op_stats[1]]
    loaded = [l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l or "libpnode_amd" in l]
    res["rccl_loaded"] = any("librccl" in p for p in loaded)
    res["pnode_amd_loaded"] = any("libpnode_amd" in p for p in loaded)
    print(json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
