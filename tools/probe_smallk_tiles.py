"""The GEMM core's tile shapes on the Cholesky's K = 384 updates, alone on the chip: C[M x N] -= A B^T (beta = 1, fp64 atomics) with
64 x 64, 64 x 128 and 128 x 128 block tiles at the shapes of a trailing update of n = 8192 (FAR: 7040 x 7040 rect stand-in, NEAR: 7040 x 384).
    python tools/probe_smallk_tiles.py [K]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops
from pygpr_amd._lib import GEMM_NT, GEMM_NT_64, GEMM_NT_64x128
ops = get_ops()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 384
g = torch.Generator(device="cuda").manual_seed(1)
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(4): fn()
        b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b) / 4)
    return best
for (M, N) in ((7040, 3584), (7040, 384), (3584, 3584), (3584, 384), (7040, 7040)):
    a = torch.randn(M, K, device="cuda", dtype=torch.float64, generator=g)
    b = torch.randn(N, K, device="cuda", dtype=torch.float64, generator=g)
    c = torch.zeros(M, N, device="cuda", dtype=torch.float64)
    for name, var in (("64x64", GEMM_NT_64), ("64x128", GEMM_NT_64x128), ("128x128", GEMM_NT)):
        t = ev(lambda: ops.gemm_raw(var, M, N, K, -1.0, a, b, 1.0, c))
        print(f"M={M} N={N} K={K} {name}: {t*1e3:.0f} us  {2.0*M*N*K/t/1e9:.1f} TFLOP/s", flush=True)
