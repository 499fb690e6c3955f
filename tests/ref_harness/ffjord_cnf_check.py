"""python ffjord_cnf_check.py -- the reference's vendored FFJORD continuous-normalising-flow layer
(/root/reference/ffjord-pnode/lib/layers/cnf.py, which wraps ODEPetsc: a 1-D flattened tuple state (z, logp), a NEW
FlattenFunc object on every forward, a func that calls autograd inside its forward for the divergence) imported from where
it lies and run UNMODIFIED against the package, then checked against autograd through the unrolled RK steps of the same
flattened func (oracle/autograd_rk.py).  Prints one JSON line with the relative differences."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pnode_amd_ref_plugin  # noqa: F401,E402  (petsc4py shim, pnode shim, CPU stand-in for the device entry points)
import petsc4py  # noqa: E402

petsc4py.init(["prog", "-ts_adapt_type", "none"])
sys.path.insert(0, "/root/reference/ffjord-pnode")
import torch  # noqa: E402
import lib.layers as layers  # noqa: E402
from lib.layers.cnf import FlattenFunc, _flatten  # noqa: E402
from oracle.autograd_rk import odeint_unrolled  # noqa: E402


def build(method):
    torch.manual_seed(0)
    diffeq = layers.ODEnet(hidden_dims=(16, 16), input_shape=(2,), strides=None, conv=False, layer_type="concat", nonlinearity="tanh")
    odefunc = layers.ODEfunc(diffeq=diffeq, divergence_fn="brute_force", residual=False, rademacher=False)
    cnf = layers.CNF(odefunc=odefunc, T=0.5, train_T=False, regularization_fns=None, solver=method)
    cnf.solver_options["step_size"] = 0.05
    return cnf


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item()


res = {}
for method, rk in (("rk4", "4"), ("dopri5_fixed", "3bs")):          # "dopri5_fixed" (cnf.py:28) is not in the method map: PETSc's default 3bs
    torch.manual_seed(1)
    x = torch.randn(7, 2)
    cnf = build(method)
    z, dlogp = cnf(x, torch.zeros(7, 1))
    ((z ** 2).mean() + dlogp.mean()).backward()
    g = torch.cat([p.grad.reshape(-1) for p in cnf.parameters() if p.grad is not None])
    ref = build(method)
    ref.odefunc.before_odeint()
    states = (x, torch.zeros(7, 1))
    u0 = _flatten(states)
    f = FlattenFunc(ref.odefunc, states)
    n = cnf.ode.num_steps
    out = odeint_unrolled(f, u0, [0.05 * (k + 1) for k in range(n)], [0.05] * n, [n], method=rk)[-1]
    z2, l2 = out[:14].view(7, 2), out[14:].view(7, 1)
    ((z2 ** 2).mean() + l2.mean()).backward()
    g2 = torch.cat([p.grad.reshape(-1) for p in ref.parameters() if p.grad is not None])
    res[method] = {"steps": n, "z": rel(z, z2), "dlogp": rel(dlogp, l2), "grad": rel(g, g2), "grad_norm": g2.norm().item()}
print(json.dumps(res))
