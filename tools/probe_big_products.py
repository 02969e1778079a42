"""The large products of one N = 16384 evaluation on blocks of matrices with leading dimension 16384 + pad: the four h = 8192 products
of the recursive split (panel product, trailing update, the two of the triangular inverse's top level) and L^-T L^-1.
TFLOP/s over the tile-granular flop the launch executes."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd import _lib
from pygpr_amd._ops import get_ops
from pygpr_amd._lib import GEMM_NT, GEMM_NN, GEMM_TN, GEMM_TT
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
pads = [int(a) for a in sys.argv[1:]] or [0, 16, 144]
for pad in pads:
    ld = N + pad
    A = torch.randn(N, ld, device="cuda", dtype=torch.float64) * 1e-3
    M = torch.randn(N, ld, device="cuda", dtype=torch.float64) * 1e-3
    a, m = A.data_ptr(), M.data_ptr()
    off = lambda r, c: (r * ld + c) * 8
    t = h // 128
    tri_tiles = t * (t + 1) // 2
    fl_tri_k = 2.0 * 128 * 128 * 128 * sum((j + 1) for j in range(t)) * t        # K range grows with the tile row / column, t tiles across
    res = []
    res.append(("panel L21 = K21 M11^T (NT khi=2)", fl_tri_k, ev(lambda: raw(GEMM_NT, h, h, h, 1.0, m + off(h, 0), ld, m, ld, 0.0, a + off(h, 0), ld, 0, 0, 2))))
    res.append(("update A22 -= L21 L21^T (NT tri)", 2.0 * 128 * 128 * h * tri_tiles, ev(lambda: raw(GEMM_NT, h, h, h, -1.0, a + off(h, 0), ld, a + off(h, 0), ld, 1.0, a + off(h, h), ld, 1, 0, 0))))
    res.append(("S = (L21 M11)^T (TT klo=1)", fl_tri_k, ev(lambda: raw(GEMM_TT, h, h, h, 1.0, m, ld, a + off(h, 0), ld, 0.0, m + off(0, h), ld, 0, 1, 0))))
    res.append(("S, K walked from its end (klo=1, krev)", fl_tri_k, ev(lambda: raw(GEMM_TT, h, h, h, 1.0, m, ld, a + off(h, 0), ld, 0.0, m + off(0, h), ld, 0, 5, 0))))
    res.append(("M21 = -M22 S^T (NT khi=1)", fl_tri_k, ev(lambda: raw(GEMM_NT, h, h, h, -1.0, m + off(h, h), ld, m + off(0, h), ld, 0.0, m + off(h, 0), ld, 0, 0, 1))))
    T = N // 128
    fl_lauum = 2.0 * 128 * 128 * 128 * sum((T - i) * (i + 1) for i in range(T))
    res.append(("K^-1 = M^T M (TN tri klo=1, n = 16384)", fl_lauum, ev(lambda: raw(GEMM_TN, N, N, N, 1.0, m, ld, m, ld, 0.0, a, ld, 1, 1, 0))))
    res.append(("K^-1, K walked from its end + columns dealt to XCDs (krev)", fl_lauum, ev(lambda: raw(GEMM_TN, N, N, N, 1.0, m, ld, m, ld, 0.0, a, ld, 1, 5, 0))))
    print(f"ld = {ld}:", flush=True)
    for name, fl, ms in res:
        print(f"   {name:44s} {ms:7.3f} ms  {fl / ms / 1e9:5.1f} TFLOP/s", flush=True)
    del A, M
