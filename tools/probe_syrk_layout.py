"""The trailing update C -= P P^T (lower tiles, K = 1024) as the factorisation issues it -- P and C are blocks of ONE matrix with
leading dimension n -- against the same product on separately allocated operands, and against a padded leading dimension."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd import _lib
from pygpr_amd._ops import get_ops, _p
from pygpr_amd._lib import GEMM_NT
ops = get_ops()
def raw(m, n, k, a, lda, b, ldb, c, ldc):
    _lib.check(ops.lib.pg_gemm_raw(ops.h, _lib.PG_F64, GEMM_NT, m, n, k, -1.0, C.c_void_p(a), lda, C.c_void_p(b), ldb, 1.0, C.c_void_p(c), ldc,
                                   1, 0, 0, ops._st()), "gemm_raw")
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
N, K = 16384, 1024
for pad in (0, 16, 64, 128, 144, 272):
    ld = N + pad
    big = torch.randn(N, ld, device="cuda", dtype=torch.float64)
    base = big.data_ptr()
    for o2 in (2048, 8192):
        m = N - o2
        pa = base + (o2 * ld + 0) * 8          # P = big[o2:, 0:K]
        pc = base + (o2 * ld + o2) * 8         # C = big[o2:, o2:]
        t = ev(lambda: raw(m, m, K, pa, ld, pa, ld, pc, ld))
        print(f"in situ   ld={ld} m={m} K={K}: {t:.3f} ms {m*(m+128)*K/t/1e9:.1f} TF/s", flush=True)
    del big
for m in (14336, 8192):
    p = torch.randn(m, K, device="cuda", dtype=torch.float64); c = torch.randn(m, m, device="cuda", dtype=torch.float64)
    t = ev(lambda: raw(m, m, K, p.data_ptr(), K, p.data_ptr(), K, c.data_ptr(), m))
    print(f"separate  m={m} K={K}: {t:.3f} ms {m*(m+128)*K/t/1e9:.1f} TF/s", flush=True)
