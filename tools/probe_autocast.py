"""Development probe: a solve under torch.autocast(bfloat16) -- default options against -pn_linear_param_grads 0."""
import os, sys, warnings
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import pnode_amd
from pnode_amd import options, petsc_adjoint
from problems import MLPFunc

dev = torch.device("cuda:0")


class Cast(torch.nn.Module):
    """func's output back in the state's precision (the engine, like the reference's PETSc vectors, takes nothing else)."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, t, y):
        return self.net(t, y).float()


def run(opts, calls=4):
    options.clear()
    for k, v in opts.items():
        options.set_option(k, v)
    torch.manual_seed(0)
    f = Cast(MLPFunc(128, torch.float32)).to(dev)
    y0 = torch.randn(512, 128, device=dev)
    o = petsc_adjoint.ODEPetsc()
    o.setupTS(y0, f, step_size=0.05, method="rk4")
    options.clear()
    res = None
    with warnings.catch_warnings(record=True) as c:
        warnings.simplefilter("always")
        for _ in range(calls):
            for p in f.parameters():
                p.grad = None
            y = y0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = o.odeint_adjoint(y, torch.tensor([0.3]))
            out.float().abs().mean().backward()
            res = (out.detach().float().clone(), y.grad.clone(), torch.cat([p.grad.reshape(-1) for p in f.parameters()]))
    return res, o, [str(m.message)[:200] for m in c if "pnode_amd" in str(m.message)]


base = {"ts_adapt_type": "none"}
ref, o_r, _ = run(dict(base, pn_linear_param_grads=0, pn_graph_capture=0))
for name, opts in (("eager", dict(base, pn_graph_capture=0)), ("default", base)):
    got, o, msgs = run(opts)
    d = [float((a - b).norm() / b.norm()) for a, b in zip(got, ref)]
    print("%-8s rel diff vs autograd's parameter gradients: out %.1e dy0 %.1e dtheta %.1e | %s | %s | %s" % (name, d[0], d[1], d[2], o.linear_param_grads[:60], o.graph_status[:40], msgs[:1]))
