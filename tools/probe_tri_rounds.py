"""Lower-tile NT launches of equally long tiles (K = 8192) against the tile count: is the time a staircase in rounds of 512 slots?"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd import _lib
from pygpr_amd._ops import get_ops
from pygpr_amd._lib import GEMM_NT
ops = get_ops()
def raw(var, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, tri=0, klo=0, khi=0):
    _lib.check(ops.lib.pg_gemm_raw(ops.h, _lib.PG_F64, var, m, n, k, float(alpha), C.c_void_p(a), lda, C.c_void_p(b), ldb, float(beta), C.c_void_p(c), ldc,
                                   tri, klo, khi, ops._st()), "gemm_raw")
def ev(fn, reps=3):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
N = 16384
A = torch.randn(N, N, device="cuda", dtype=torch.float64) * 1e-3
a = A.data_ptr()
off = lambda r, c: (r * N + c) * 8
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
for t in [int(x) for x in sys.argv[2:]] or [30, 31, 32, 44, 45, 46, 54, 55, 56, 60, 62, 63, 64, 66]:
    h = 128 * t
    tiles = t * (t + 1) // 2
    ms = ev(lambda: raw(GEMM_NT, h, h, K, -1.0, a + off(N - h, 0), N, a + off(N - h, 0), N, 1.0, a + off(N - h, N - h), N, 1))
    print(f"t={t} tiles={tiles} rounds={tiles/512:.2f} K={K}: {ms:.3f} ms  {2.0*128*128*K*tiles/ms/1e9:5.1f} TFLOP/s  per round-up {ms/-(-tiles//512):.3f} ms", flush=True)
