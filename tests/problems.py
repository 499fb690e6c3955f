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


def flat_grads(module):
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                      for p in module.parameters() if p.requires_grad])


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()
