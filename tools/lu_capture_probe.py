import torch, gc
dev = torch.device("cuda:0")
torch.manual_seed(0)
n = 1024
A = torch.randn(n, n, dtype=torch.float64, device=dev) + n ** 0.5 * torch.eye(n, dtype=torch.float64, device=dev)
R = torch.randn(64, n, dtype=torch.float64, device=dev)
LU, piv = torch.linalg.lu_factor(A)
X0 = torch.linalg.lu_solve(LU, piv, R, left=False)
X1 = torch.linalg.lu_solve(LU, piv, R, left=False, adjoint=True)
torch.cuda.synchronize()
static = R.clone()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        Y0 = torch.linalg.lu_solve(LU, piv, static, left=False)
        Y1 = torch.linalg.lu_solve(LU, piv, static, left=False, adjoint=True)
    for i in range(3):
        static.copy_(R * (i + 1)); g.replay(); torch.cuda.current_stream().synchronize()
        print("replay", i, torch.equal(Y0, X0 * (i + 1)) or ((Y0 - X0 * (i + 1)).norm() / X0.norm()).item(), torch.equal(Y1, X1 * (i + 1)) or ((Y1 - X1 * (i + 1)).norm() / X1.norm()).item())
except Exception as e:
    print("capture failed:", repr(e)[:300])
# in-place refactor into the same tensors
try:
    A2 = A + 0.1 * torch.eye(n, dtype=torch.float64, device=dev)
    torch.linalg.lu_factor_ex(A2, check_errors=False, out=(LU, piv, torch.empty((), dtype=torch.int32, device=dev)))
    g.replay(); torch.cuda.synchronize()
    ref = torch.linalg.lu_solve(*torch.linalg.lu_factor(A2), static, left=False)
    print("after in-place refactor: rel", ((Y0 - ref).norm() / ref.norm()).item())
except Exception as e:
    print("refactor failed:", repr(e)[:300])
