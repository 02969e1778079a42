"""One rows kernel of the coupled chain on its own (flags preset: nothing to wait for) against the GEMM core's skinny update at the same shape:
how efficient is the window's product inside the rows kernel?  m rows below the tile, window of K columns."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops
from pygpr_amd import _lib
ops = get_ops()
def ev(fn, reps=10):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b) / reps)
    return best * 1e3
g = torch.Generator(device="cuda").manual_seed(1)
n = 8192
A = torch.randn(n, n, device="cuda", dtype=torch.float64, generator=g) * 0.01
inv = torch.tril(torch.randn(128, 128, device="cuda", dtype=torch.float64, generator=g)).contiguous() * 0.05
flags = torch.zeros(16, dtype=torch.int32, device="cuda"); info = torch.zeros(1, dtype=torch.int32, device="cuda")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def rows(o0, k0):
    _lib.check(ops.lib.pg_rowstep_raw(ops.h, 0, n, C.c_void_p(A.data_ptr()), A.stride(0), o0, k0, C.c_void_p(inv.data_ptr()),
                                      C.c_void_p(flags.data_ptr()), C.c_void_p(info.data_ptr()), st()), "pg_rowstep_raw")
def gemm(variant, m, nn, k, a, b, c):
    _lib.check(ops.lib.pg_gemm_raw(ops.h, 0, variant, m, nn, k, -1.0, C.c_void_p(a.data_ptr()), a.stride(0), C.c_void_p(b.data_ptr()), b.stride(0),
                                   1.0, C.c_void_p(c.data_ptr()), c.stride(0), 0, 0, 0, st()), "pg_gemm_raw")
for m in (2048, 4096, 6144):
    k0 = n - 128 - m
    for K in (128, 384, 640):
        if k0 - K < 0: continue
        t = ev(lambda: rows(k0 - K, k0))
        flop = m * 128 * (K + 64 + 128) * 2.0            # window + solve against the triangular inverse + the last 128 columns
        a = A[k0 + 128:k0 + 128 + m, k0 - K:k0]; b = A[k0 + 128:k0 + 256, k0 - K:k0]; c = A[k0 + 128:k0 + 128 + m, k0 + 128:k0 + 256]
        tg = {}
        for name, v in (("64x64", 6), ("32x64", 8), ("64x128", 7)):
            tg[name] = ev(lambda: gemm(v, m, 128, K + 128, a, b, c))
        print(f"m={m} window K={K}: rows kernel {t:6.1f} us = {flop/t/1e6:5.1f} TF/s | GEMM core, M x 128 x (K+128): " +
              "  ".join(f"{k} {v:6.1f} us ({m*128*(K+128)*2/v/1e6:4.1f} TF/s)" for k, v in tg.items()), flush=True)
