"""Uniform products on the GEMM core in fp32 (the config-5 regime): rate per instantiation."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops
from pygpr_amd._lib import GEMM_NT, GEMM_NN, GEMM_TN, GEMM_TT
ops = get_ops()
def ev(fn, reps=4):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
g = torch.Generator(device="cuda").manual_seed(1)
for dt, peak in ((torch.float32, 157.3), (torch.float64, 78.6)):
    n = 8192
    a = torch.randn(n, n, device="cuda", dtype=dt, generator=g)
    b = torch.randn(n, n, device="cuda", dtype=dt, generator=g)
    c = torch.zeros(n, n, device="cuda", dtype=dt)
    for name, var in (("NT", GEMM_NT), ("NN", GEMM_NN), ("TN", GEMM_TN), ("TT", GEMM_TT)):
        t = ev(lambda: ops.gemm_raw(var, n, n, n, 1.0, a, b, 0.0, c))
        print(f"{dt} uniform {name} {n}^3: {t:.3f} ms {2*n**3/t/1e9:.1f} TF/s ({2*n**3/t/1e9/peak*100:.0f} % of peak)", flush=True)
