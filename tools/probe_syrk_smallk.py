"""Where does a short-K lower SYRK lose its time?  Same launch with the atomic epilogue (beta = 1), the plain store epilogue (beta = 0), and
both tile shapes, at the shapes the factorisation's trailing update uses."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops
from pygpr_amd._lib import GEMM_NT, GEMM_NT_64
ops = get_ops()
def ev(fn, reps=7):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
g = torch.Generator(device="cuda").manual_seed(1)
for n in (7424, 16384 - 2048):
    c = torch.randn(n, n, device="cuda", dtype=torch.float64, generator=g)
    for K in (128, 256, 384, 512, 768, 1024, 2048):
        p = torch.randn(n, K, device="cuda", dtype=torch.float64, generator=g)
        out = []
        for name, var in (("128", GEMM_NT), ("64", GEMM_NT_64)):
            for beta in (1.0, 0.0):
                t = ev(lambda: ops.gemm_raw(var, n, n, K, -1.0, p, p, beta, c, tri=1))
                tile = 128 if var == GEMM_NT else 64
                out.append(f"t{name} beta={beta:.0f}: {t*1e3:7.1f} us {n*(n+tile)*K/t/1e9:5.1f} TF")
        print(f"n={n} K={K}: " + " | ".join(out), flush=True)
    del c
