"""Prototype measurement (round 2): a 256 x 128 block with sixteen waves as variant 13 of pg_gemm_raw -- the variant is NOT in the
library any more (66.0 vs 71.5 TFLOP/s on a uniform 8192^3 product); kept for the record of how it was measured."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops
ops = get_ops()
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
g = torch.Generator(device="cuda").manual_seed(1)
nu = 8192
a = torch.randn(nu, nu, device="cuda", dtype=torch.float64, generator=g)
b = torch.randn(nu, nu, device="cuda", dtype=torch.float64, generator=g)
c0 = torch.zeros(nu, nu, device="cuda", dtype=torch.float64)
c1 = torch.zeros(nu, nu, device="cuda", dtype=torch.float64)
for var in (0, 13):
    c = c0 if var == 0 else c1
    t = ev(lambda: ops.gemm_raw(var, nu, nu, nu, 1.0, a, b, 0.0, c))
    print(f"uniform NT variant {var}: {t:.3f} ms {2*nu**3/t/1e9:.1f} TF/s", flush=True)
print("max diff", float((c0 - c1).abs().max()))
for K in (1024, 2048):
    p = torch.randn(14336, K, device="cuda", dtype=torch.float64, generator=g)
    q = torch.randn(1024, K, device="cuda", dtype=torch.float64, generator=g)
    cc = torch.zeros(14336, 1024, device="cuda", dtype=torch.float64)
    for var in (0, 13):
        t = ev(lambda: ops.gemm_raw(var, 14336, 1024, K, -1.0, p, q, 1.0, cc))
        print(f"Sa-like 14336x1024xK={K} variant {var}: {t:.3f} ms {2*14336*1024*K/t/1e9:.1f} TF/s", flush=True)
