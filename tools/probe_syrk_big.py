"""Why does the K = 8192 trailing update of the recursive split run at 61 TFLOP/s when its neighbours reach 68-70?  Variants of
C (-)= P P^T on h = 8192: lower tiles / full square, beta = 1 (atomic epilogue) / 0, K = 4096 / 8192, separate operands."""
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
N, h = 16384, 8192
A = torch.randn(N, N, device="cuda", dtype=torch.float64) * 1e-3
B = torch.randn(h, h, device="cuda", dtype=torch.float64) * 1e-3
a, b = A.data_ptr(), B.data_ptr()
off = lambda r, c: (r * N + c) * 8
t = h // 128
tri_tiles = t * (t + 1) // 2
def show(name, fl, ms): print(f"{name:60s} {ms:7.3f} ms  {fl / ms / 1e9:5.1f} TFLOP/s", flush=True)
for K in (4096, 8192):
    ftri, ffull = 2.0 * 128 * 128 * K * tri_tiles, 2.0 * h * h * K
    show(f"in situ tri  beta=1 K={K}", ftri, ev(lambda: raw(GEMM_NT, h, h, K, -1.0, a + off(h, 0), N, a + off(h, 0), N, 1.0, a + off(h, h), N, 1)))
    show(f"in situ tri  beta=0 K={K}", ftri, ev(lambda: raw(GEMM_NT, h, h, K, -1.0, a + off(h, 0), N, a + off(h, 0), N, 0.0, a + off(h, h), N, 1)))
    show(f"in situ full beta=0 K={K} (B = A)", ffull, ev(lambda: raw(GEMM_NT, h, h, K, -1.0, a + off(h, 0), N, a + off(h, 0), N, 0.0, a + off(h, h), N, 0)))
    show(f"in situ full beta=0 K={K} (B = other rows)", ffull, ev(lambda: raw(GEMM_NT, h, h, K, -1.0, a + off(h, 0), N, a + off(0, 0), N, 0.0, a + off(h, h), N, 0)))
    show(f"separate B[h,h] tri beta=0 K={K} -> C in A", ftri, ev(lambda: raw(GEMM_NT, h, h, K, -1.0, b, h, b, h, 0.0, a + off(h, h), N, 1)))
