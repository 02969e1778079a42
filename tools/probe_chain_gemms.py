"""The two skinny products of the Cholesky chain at the shapes it issues (back-to-back on one stream):
U: C[M,128] -= A[M,K] B[128,K]^T (64x64 tiles), T: B[M,128] <- B inv^T (64x128 tiles, K = 128)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops
from pygpr_amd._lib import GEMM_NT_64
ops = get_ops()
GEMM_NT_64x128 = 7
import ctypes as C
from pygpr_amd import _lib
def raw(variant, m, n, k, alpha, a, b, beta, c, khi=0):   # strided views, as the driver passes them
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(ops.lib.pg_gemm_raw(ops.h, 0, variant, m, n, k, float(alpha), C.c_void_p(a.data_ptr()), a.stride(0),
                                   C.c_void_p(b.data_ptr()), b.stride(0), float(beta), C.c_void_p(c.data_ptr()), c.stride(0),
                                   0, 0, khi, st), "pg_gemm_raw")
def ev(fn, reps=20):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b) / reps)
    return best * 1e3
g = torch.Generator(device="cuda").manual_seed(1)
big = torch.randn(16384, 2048, device="cuda", dtype=torch.float64, generator=g)
inv = torch.tril(torch.randn(128, 128, device="cuda", dtype=torch.float64, generator=g)).contiguous()
for M in (1024, 4096, 8192, 15360):
    a = big[:M]
    c = a[:, 1024:1152]
    for name, vu, vt in (("64-row tiles", GEMM_NT_64, 7), ("32-row tiles", 8, 9)):
        line = [f"M={M:6d} {name}"]
        for K in (128, 384, 896):
            t = ev(lambda: raw(vu, M, 128, K, -1.0, a, big[:128], 1.0, c))
            line.append(f"U K={K}: {t:6.1f} us")
        t = ev(lambda: raw(vt, M, 128, 128, 1.0, c, inv, 0.0, c, khi=2))
        line.append(f"T: {t:6.1f} us")
        print("  ".join(line), flush=True)
# the 32-row variants against the 64-row ones
M = 4096
a = big[:M]
c1 = torch.randn(M, 128, device="cuda", dtype=torch.float64, generator=g); c2 = c1.clone()
raw(GEMM_NT_64, M, 128, 384, -1.0, a, big[:128], 1.0, c1); raw(8, M, 128, 384, -1.0, a, big[:128], 1.0, c2)
print("U max diff", float((c1 - c2).abs().max()))
c1 = torch.randn(M, 128, device="cuda", dtype=torch.float64, generator=g); c2 = c1.clone()
raw(7, M, 128, 128, 1.0, c1, inv, 0.0, c1, khi=2); raw(9, M, 128, 128, 1.0, c2, inv, 0.0, c2, khi=2)
print("T max diff", float((c1 - c2).abs().max()), float(c1.abs().max()))
