import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops
from pygpr_amd._lib import GEMM_NT, GEMM_TN
ops = get_ops()
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
n = 16384
g = torch.Generator(device="cuda").manual_seed(1)
c = torch.randn(n, n, device="cuda", dtype=torch.float64, generator=g)
for K in (256, 1024, 4096):
    p = torch.randn(n, K, device="cuda", dtype=torch.float64, generator=g)
    t = ev(lambda: ops.gemm_raw(GEMM_NT, n, n, K, -1.0, p, p, 1.0, c, tri=1))
    print(f"syrk lower n={n} K={K}: {t:.3f} ms {n*(n+128)*K/t/1e9:.1f} TF/s", flush=True)
m = torch.tril(torch.randn(n, n, device="cuda", dtype=torch.float64, generator=g))
t = ev(lambda: ops.gemm_raw(GEMM_TN, n, n, n, 1.0, m, m, 0.0, c, tri=1, klo=1), 3)
print(f"lauum n={n}: {t:.3f} ms {n**3/3/t/1e9:.1f} TF/s")
del m, c
nu = 8192
a = torch.randn(nu, nu, device="cuda", dtype=torch.float64, generator=g)
b = torch.randn(nu, nu, device="cuda", dtype=torch.float64, generator=g)
cu = torch.zeros(nu, nu, device="cuda", dtype=torch.float64)
from pygpr_amd._lib import GEMM_NN, GEMM_TT
for name, var in (("NT", GEMM_NT), ("NN", GEMM_NN), ("TN", GEMM_TN), ("TT", GEMM_TT)):
    t = ev(lambda: ops.gemm_raw(var, nu, nu, nu, 1.0, a, b, 0.0, cu))
    print(f"uniform {name} {nu}^3: {t:.3f} ms {2*nu**3/t/1e9:.1f} TF/s", flush=True)
