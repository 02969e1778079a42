"""Does a non-power-of-two leading dimension change the GEMM core's rate?  Uniform 8192^3 products on views of wider
buffers (ld = n + pad), through the raw entry point with explicit strides."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops
from pygpr_amd import _lib
from pygpr_amd._lib import GEMM_NT, GEMM_NN, GEMM_TN, GEMM_TT
ops = get_ops()
def raw(variant, m, n, k, alpha, a, b, beta, c, tri=0, klo=0, khi=0):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(ops.lib.pg_gemm_raw(ops.h, 0, variant, m, n, k, float(alpha), C.c_void_p(a.data_ptr()), a.stride(0),
                                   C.c_void_p(b.data_ptr()), b.stride(0), float(beta), C.c_void_p(c.data_ptr()), c.stride(0),
                                   tri, klo, khi, st), "pg_gemm_raw")
def ev(fn, reps=4):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
g = torch.Generator(device="cuda").manual_seed(1)
n = 8192
for pad in (0, 32, 64, 256, 288):
    ld = n + pad
    A = torch.randn(n, ld, device="cuda", dtype=torch.float64, generator=g)[:, :n]
    B = torch.randn(n, ld, device="cuda", dtype=torch.float64, generator=g)[:, :n]
    Cm = torch.zeros(n, ld, device="cuda", dtype=torch.float64)[:, :n]
    line = [f"ld = n + {pad:3d}:"]
    for name, var in (("NT", GEMM_NT), ("NN", GEMM_NN), ("TN", GEMM_TN), ("TT", GEMM_TT)):
        t = ev(lambda: raw(var, n, n, n, 1.0, A, B, 0.0, Cm))
        line.append(f"{name} {2*n**3/t/1e9:5.1f}")
    t = ev(lambda: raw(GEMM_TN, n, n, n, 1.0, A, A, 0.0, Cm, tri=1, klo=1))
    line.append(f"lauum-shaped {n**3/3/t/1e9:5.1f}")
    t = ev(lambda: raw(GEMM_NT, n, n, 1024, -1.0, A, A, 1.0, Cm, tri=1))
    line.append(f"syrk K=1024 {n*(n+128)*1024/t/1e9:5.1f} TF/s")
    print("  ".join(line), flush=True)
    del A, B, Cm
