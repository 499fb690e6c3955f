"""Experiment (round 5): find the first operation of the IMEX direct-solve reverse sweep whose result differs between the
eager twin and the replayed capture of the same call.  Every lincomb / VJP / direct solve of the validating call is
recorded (clones; the captured pass's clones live in the graph's pool and hold the replay's values afterwards)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pnode_amd import options, petsc_adjoint, _sweepgraphs
from problems import DiffusionIM, ReactionEX

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "3"
_sweepgraphs.SweepGraphs.AUTO_THETA = True
for k, v in {"ts_adapt_type": "none", "ts_arkimex_type": name, "snes_type": "ksponly"}.items():
    options.set_option(k, v)
torch.manual_seed(5)
fI, fE = DiffusionIM(16).to(dev), ReactionEX(16).to(dev)
y0 = torch.randn(8, 16, dtype=torch.float64, device=dev)
t = torch.tensor([0.0, 0.1, 0.25], dtype=torch.float64)
ode = petsc_adjoint.ODEPetsc()
ode.setupTS(y0, fI, step_size=0.05, method="imex", implicit_form=True, imex_form=True, func2=fE, batch_size=8,
            linear_solver="torch", matrixfree_jacobian=False)
options.clear()
params = list(fI.parameters()) + list(fE.parameters())
rec = []
on = [False]
ops = ode._ops
orig_lincomb, orig_copy, orig_vjp = ops.lincomb, ops.copy, ode._vjp
orig_pam = ops.param_accum_multi
orig_pa = ops.param_accum


def lincomb(out, xs, cs):
    orig_lincomb(out, xs, cs)
    if on[0]:
        rec.append(("lincomb%d" % len(xs), out.clone()))


def copy(y, x):
    orig_copy(y, x)
    if on[0]:
        rec.append(("copy", y.clone()))


def vjp(*a, **k):
    gy, gp = orig_vjp(*a, **k)
    if on[0]:
        rec.append(("vjp.gy/" + k.get("which", "EX"), None if gy is None else gy.clone()))
        for j, g in enumerate(gp):
            rec.append(("vjp.gp%d/" % j + k.get("which", "EX"), None if g is None else g.clone()))
    return gy, gp


def pam(mu, alphas, sets, off, ln):
    orig_pam(mu, alphas, sets, off, ln)
    if on[0]:
        rec.append(("param_accum_multi[%d]" % len(sets), mu.clone()))


def pa(mu, alpha, grads, off, ln):
    orig_pa(mu, alpha, grads, off, ln)
    if on[0]:
        rec.append(("param_accum", mu.clone()))


ops.lincomb, ops.copy, ode._vjp, ops.param_accum_multi, ops.param_accum = lincomb, copy, vjp, pam, pa
orig_rev = ode._reverse_sweep


def rev(g, T):
    rec.append(("== reverse sweep", None))
    return orig_rev(g, T)


ode._reverse_sweep = rev
for it in range(3):
    for p in params:
        p.grad = None
    on[0] = it == 2
    yin = (y0 * (1.0 + 0.1 * it)).requires_grad_(True)
    sol = ode.odeint_adjoint(yin, t.to(dev))
    sol.abs().mean().backward()
torch.cuda.synchronize()
print(ode.graph_status)
marks = [i for i, (n, _) in enumerate(rec) if n.startswith("==")]
print("records", len(rec), "reverse markers at", marks)
a, b = rec[marks[-2] + 1:marks[-1]], rec[marks[-1] + 1:]
print("eager reverse ops", len(a), "captured reverse ops", len(b))
shown = 0
for i, ((na, xa), (nb, xb)) in enumerate(zip(a, b)):
    same = (xa is None and xb is None) or (xa is not None and xb is not None and torch.equal(xa, xb))
    if na != nb or not same:
        d = float((xa - xb).abs().max()) if (xa is not None and xb is not None and xa.shape == xb.shape) else None
        print("op %d: %s vs %s  equal=%s maxabs=%s ptr%%256 %s/%s" % (i, na, nb, same, d, xa.data_ptr() % 256 if xa is not None else None,
                                                                     xb.data_ptr() % 256 if xb is not None else None))
        shown += 1
        if shown > 12:
            break
print("first 40 op names:", [n for n, _ in a[:40]])
# ---- which side deviates?  redo the first diverging direct solve eagerly with both sets of factors
th = ode._theta
print("factor keys eager", list(th._lu), "static", list(th._static_lu))
for key in th._lu:
    LUe, pive = th._lu[key][:2]
    LUs, pivs = th._static_lu[key][:2]
    print("key", key, "LU equal", bool(torch.equal(LUe, LUs)), "maxabs", float((LUe - LUs).abs().max()), "piv equal", bool(torch.equal(pive, pivs)))
idx = next(i for i, ((na, xa), (nb, xb)) in enumerate(zip(a, b)) if xa is not None and not torch.equal(xa, xb))
rhs = a[idx - 1][1]
assert torch.equal(rhs, b[idx - 1][1])
n1 = 16
R = rhs[: ode.n].view(-1, n1)
for key in th._lu:
    for tag, (LU, piv) in (("eager factors", th._lu[key][:2]), ("static factors", th._static_lu[key][:2])):
        X = torch.linalg.lu_solve(LU, piv, R, left=False, adjoint=True).contiguous().reshape(-1)
        print(key, tag, "== eager twin's result:", bool(torch.equal(X, a[idx][1][: ode.n])), " == replay's result:", bool(torch.equal(X, b[idx][1][: ode.n])))
