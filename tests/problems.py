"""Shared synthetic problems for the parity tests (shapes follow the reference's examples)."""
import torch
import torch.nn as nn


class SpiralFunc(nn.Module):
    """Linear(2,50)-Tanh-Linear(50,2) on y**3, weights N(0,0.1), zero bias
    (reference: examples-pnode/ode_demo_petsc.py:207-230)."""

    def __init__(self, dtype=torch.float64, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.net = nn.Sequential(nn.Linear(2, 50), nn.Tanh(), nn.Linear(50, 2)).to(dtype)
        for m in self.net.modules():
            if isinstance(m, nn.Linear):
                with torch.no_grad():
                    m.weight.copy_((torch.randn(m.weight.shape, generator=g, dtype=torch.float64) * 0.1).to(dtype))
                    m.bias.zero_()
        self.nfe = 0

    def forward(self, t, y):
        self.nfe += 1
        return self.net(y ** 3)


class SpiralTruth(nn.Module):
    """y' = y**3 A (ode_demo_petsc.py:83,91-93) -- stiff enough to make dopri5 adapt."""

    def __init__(self, dtype=torch.float64):
        super().__init__()
        self.A = nn.Parameter(torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], dtype=dtype))

    def forward(self, t, y):
        return torch.mm(y.reshape(-1, 2) ** 3, self.A).reshape(y.shape)


class TimeDependent(nn.Module):
    """Uses t explicitly so that stage times are checked; has an unused parameter (None grad)."""

    def __init__(self, d, dtype=torch.float64, seed=1):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        # drawn in fp64 and cast, so every dtype gets the same model
        self.W = nn.Parameter((torch.randn(d, d, generator=g, dtype=torch.float64) * 0.3).to(dtype))
        self.unused = nn.Parameter(torch.ones(3, dtype=dtype))
        self.bias = nn.Parameter((torch.randn(d, generator=g, dtype=torch.float64) * 0.1).to(dtype))

    def forward(self, t, y):
        return torch.tanh(y @ self.W) * (1.0 + 0.5 * t) + self.bias * torch.sin(torch.as_tensor(t, dtype=y.dtype))


class MLPFunc(nn.Module):
    """3x[Linear(d,d)+Tanh] + Linear(d,d), W~N(0,0.02), b=0."""

    def __init__(self, d=512, dtype=torch.float32, seed=0, std=0.02):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        layers = []
        for k in range(4):
            lin = nn.Linear(d, d)
            with torch.no_grad():
                lin.weight.copy_(torch.randn(d, d, generator=g) * std)
                lin.bias.zero_()
            layers.append(lin)
            if k < 3:
                layers.append(nn.Tanh())
        self.net = nn.Sequential(*layers).to(dtype)

    def forward(self, t, y):
        return self.net(y)


class TimeGatedMLPFunc(MLPFunc):
    """MLPFunc in which a weight or bias of one Linear layer is ALSO used outside that layer's call, at some times only
    (tools/fuzz_modes.py draws `kind` and the gate): `late` -- downstream functional use of layer 2's
    weight; `upstream` -- functional use of layer 2's weight and bias in FRONT of layer 2's own call; `bias` -- the last bias
    added a second time.  The reference differentiates every evaluation with respect to every parameter (pa.py:66-74)."""

    def __init__(self, d=512, dtype=torch.float32, seed=0, std=0.02, kind="late", gate=0.15, below=True):
        super().__init__(d, dtype, seed, std)
        self.kind, self.gate, self.below = kind, gate, below

    def forward(self, t, y):
        net = self.net
        on = (t < self.gate) == self.below
        h = net[1](net[0](y))
        if on and self.kind == "upstream":
            h = h + 0.3 * torch.tanh(torch.nn.functional.linear(h, net[2].weight, net[2].bias))
        z = net[3](net[2](h))
        if on and self.kind == "late":
            z = z + 0.5 * torch.nn.functional.linear(h, net[2].weight)
        out = net[6](net[5](net[4](z)))
        if on and self.kind == "bias":
            out = out + net[6].bias * (1.0 + t)
        return out


class SwitchedMLPFunc(MLPFunc):
    """The C3 dynamics made to ADAPT: same layers, shapes and parameter count as MLPFunc, weights
    W ~ N(0, 0.08) instead of N(0, 0.02), and the whole right-hand side gated by g(t) = tanh(20 (sin(15 pi t) + 1/2)):
    7.5 times per time unit the vector field reverses within ~0.01 time units, runs backwards for a third of the period and
    reverses again.  dopri5 at PETSc's default tolerances (integrate to T = 4) then takes well over 100 accepted steps and
    rejects a few attempts at every reversal -- the BASELINE weights give 3 steps and no rejection.  The gate is
    asymmetric so that the state makes net progress: with a symmetric one dL/dtheta is the small difference of forward
    and backward contributions and its fp32 round-off is amplified ten-fold (measured, NOTES_r04.md)."""

    T_END = 4.0

    def __init__(self, d=512, dtype=torch.float32, seed=0, std=0.08, sharp=20.0, freq=7.5, offset=0.5):
        super().__init__(d, dtype, seed, std)
        self.sharp, self.freq, self.offset = sharp, freq, offset

    def forward(self, t, y):
        import math
        if isinstance(t, torch.Tensor) and t.is_cuda:
            # the solver's per-evaluation hipGraphs (pnode_amd/_stagegraphs.py) hand t over as a 0-dim float64 device tensor: the
            # same expression in device arithmetic (a host float would synchronise, and the solver would stay with eager launches)
            return self.net(y) * torch.tanh(self.sharp * (torch.sin((2.0 * math.pi * self.freq) * t) + self.offset))
        return self.net(y) * math.tanh(self.sharp * (math.sin(2.0 * math.pi * self.freq * float(t)) + self.offset))


class RoberIM(nn.Module):
    """Implicit part of the reference's IMEX split of ROBER (reference tests/test_pnode.py:99-111)."""

    def __init__(self):
        super().__init__()
        self.k1 = nn.Parameter(torch.tensor([0.05], dtype=torch.float64))
        self.k3 = nn.Parameter(torch.tensor([2e4], dtype=torch.float64))

    def forward(self, t, y):
        k1, k3 = self.k1[0], self.k3[0]
        f1 = -k1 * y[0] + k3 * y[1] * y[2]
        f2 = k1 * y[0] - k3 * y[1] * y[2]
        return torch.stack((f1, f2, torch.zeros_like(f1)), -1)


class RoberEX(nn.Module):
    """Explicit part (reference tests/test_pnode.py:114-124)."""

    def __init__(self):
        super().__init__()
        self.k2 = nn.Parameter(torch.tensor([4e7], dtype=torch.float64))

    def forward(self, t, y):
        k2 = self.k2[0]
        f2 = -k2 * y[1] ** 2
        return torch.stack((torch.zeros_like(f2), f2, -f2), -1)


class DiffusionIM(nn.Module):
    """Linear, batch-row-wise stiff part: circular second difference scaled by a trainable
    viscosity (the shape of examples-sinode/Burgers/Burgers.py:170-195's funcIM)."""

    def __init__(self, n, dtype=torch.float64, nu=0.05):
        super().__init__()
        self.nu = nn.Parameter(torch.tensor(nu, dtype=dtype))
        self.scale = float(n * n) / 64.0

    def forward(self, t, y):
        return self.nu * self.scale * (torch.roll(y, 1, -1) - 2.0 * y + torch.roll(y, -1, -1))


class AdvectionDiffusionIM(nn.Module):
    """Linear, batch-row-wise, NONSYMMETRIC stiff part (upwind advection + diffusion, trainable speeds): a transposed
    factor in the direct stage solve or in its adjoint would show here and not with a symmetric operator."""

    def __init__(self, n, dtype=torch.float64):
        super().__init__()
        self.c = nn.Parameter(torch.tensor(0.8, dtype=dtype))
        self.nu = nn.Parameter(torch.tensor(0.05, dtype=dtype))
        self.n = n

    def forward(self, t, y):
        n = self.n
        return (-self.c * n / 4.0 * (y - torch.roll(y, 1, -1))
                + self.nu * n * n / 64.0 * (torch.roll(y, 1, -1) - 2.0 * y + torch.roll(y, -1, -1)))


class ReactionEX(nn.Module):
    """Non-stiff nonlinear part: a small MLP on each row (Burgers.py:134-160's funcEX shape)."""

    def __init__(self, n, dtype=torch.float64, seed=2):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.l1 = nn.Linear(n, n + 2)
        self.l2 = nn.Linear(n + 2, n)
        for p in self.parameters():
            with torch.no_grad():
                p.copy_((torch.randn(p.shape, generator=g, dtype=torch.float64) * 0.3).to(p.dtype))
        self.to(dtype)

    def forward(self, t, y):
        return self.l2(torch.relu(self.l1(y))) * (1.0 + 0.1 * t)


class SemiExplicitDAE(nn.Module):
    """Index-1 DAE in the reference's mass-matrix form M u' = f(t, u) (pendulum_DAE.py's shape): three
    differential components y' = tanh(y) A + z B and two algebraic ones 0 = y C - z; M = diag(1,1,1,0,0)."""

    def __init__(self, dtype=torch.float64, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.A = nn.Parameter((torch.randn(3, 3, generator=g, dtype=torch.float64) * 0.3 - torch.eye(3, dtype=torch.float64)).to(dtype))
        self.B = nn.Parameter((torch.randn(2, 3, generator=g, dtype=torch.float64) * 0.3).to(dtype))
        self.C = nn.Parameter((torch.randn(3, 2, generator=g, dtype=torch.float64) * 0.3).to(dtype))

    def forward(self, t, u):
        y, z = u[..., :3], u[..., 3:]
        return torch.cat([torch.tanh(y) @ self.A + z @ self.B, y @ self.C - z], -1)

    def consistent(self, y):
        return torch.cat([y, y @ self.C.detach()], -1)

    @staticmethod
    def mass(dtype=torch.float64):
        return torch.diag(torch.tensor([1.0, 1.0, 1.0, 0.0, 0.0], dtype=dtype))


def flat_grads(module):
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                      for p in module.parameters() if p.requires_grad])


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


class ConvBlockFunc(nn.Module):
    """The func of BASELINE config C4, written from its layer shapes (reference
    examples-pnode/models/sqnxt_PETSc.py:70-121, ``BasicBlock2(dim)``): five Conv2d + BatchNorm2d + ReLU,
    channels dim -> dim/2 -> dim/4 -> dim/2 -> dim/2 -> dim with kernels 1x1, 1x1, 1x3, 3x1, 1x1 (same
    padding, bias on).  dim = 64: 9 744 trainable parameters.  BatchNorm runs in eval mode
    with non-trivial running statistics and affine terms, so that the func is deterministic and
    row-wise in the batch (train-mode BN couples the batch)."""

    def __init__(self, dim=64, dtype=torch.float32, seed=0):
        super().__init__()
        h, q = dim // 2, dim // 4
        spec = [(dim, h, (1, 1)), (h, q, (1, 1)), (q, h, (1, 3)), (h, h, (3, 1)), (h, dim, (1, 1))]
        g = torch.Generator().manual_seed(seed)
        self.convs = nn.ModuleList()
        self.norms = nn.ModuleList()
        for cin, cout, k in spec:
            conv = nn.Conv2d(cin, cout, k, stride=1, padding=(k[0] // 2, k[1] // 2), bias=True)
            bn = nn.BatchNorm2d(cout)
            fan_in = cin * k[0] * k[1]
            with torch.no_grad():
                conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (1.0 / fan_in) ** 0.5)
                conv.bias.copy_(torch.randn(cout, generator=g) * 0.05)
                bn.weight.copy_(1.0 + 0.1 * torch.randn(cout, generator=g))
                bn.bias.copy_(0.05 * torch.randn(cout, generator=g))
                bn.running_mean.copy_(0.1 * torch.randn(cout, generator=g))
                bn.running_var.copy_(1.0 + 0.2 * torch.rand(cout, generator=g))
            self.convs.append(conv)
            self.norms.append(bn)
        self.to(dtype)
        self.eval()
        self.nfe = 0

    def train(self, mode=True):          # the statistics stay frozen whatever the caller's model does
        return super().train(False)

    def forward(self, t, x):
        self.nfe += 1
        for conv, bn in zip(self.convs, self.norms):
            x = torch.relu(bn(conv(x)))
        return x


class BurgersIM(nn.Module):
    """Stiff part of BASELINE config C5 (reference examples-sinode/Burgers/Burgers.py:170-195 with
    fixed_linear=True): the circular three-point Laplacian alpha/dx^2 * [1, -2, 1] applied as a Conv1d
    along the spatial axis of every batch row; no trainable parameter (dx = 1/n, alpha = 8e-4)."""

    def __init__(self, n, alpha=8e-4, dtype=torch.float64):
        super().__init__()
        self.A = nn.Conv1d(1, 1, 3, padding="same", padding_mode="circular", bias=False)
        k = alpha * float(n) ** 2
        self.A.weight = nn.Parameter(torch.tensor([[[k, -2.0 * k, k]]]), requires_grad=False)
        self.to(dtype)

    def forward(self, t, y):
        return self.A(y.unsqueeze(1)).squeeze(1)


class BurgersEX(nn.Module):
    """Non-stiff part of C5 (Burgers.py:134-160): Linear n -> 9n/8, three Linear 9n/8 -> 9n/8, Linear
    9n/8 -> n with ReLU between them, zero bias.  The reference draws W ~ N(0, 0.1); at n = 1024 that makes
    an untrained net's output O(1e3) per layer, so the synthetic weights here are N(0, 1/fan_in)-scaled
    (same shapes and parameter count; the training run tames the reference's weights the same way)."""

    def __init__(self, n, dtype=torch.float64, seed=0):
        super().__init__()
        w = n * 9 // 8
        dims = [n, w, w, w, w, n]
        g = torch.Generator().manual_seed(seed)
        layers = []
        for i in range(5):
            lin = nn.Linear(dims[i], dims[i + 1])
            with torch.no_grad():
                lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) * (1.0 / dims[i]) ** 0.5)
                lin.bias.zero_()
            layers.append(lin)
            if i < 4:
                layers.append(nn.ReLU())
        self.net = nn.Sequential(*layers).to(dtype)

    def forward(self, t, y):
        return self.net(y)
