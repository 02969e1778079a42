import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops
from pygpr_amd._lib import GEMM_NT, GEMM_NN, GEMM_TN, GEMM_TT
ops = get_ops()
nu = 8192
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn(nu, nu, device="cuda", dtype=torch.float64, generator=g)
b = torch.randn(nu, nu, device="cuda", dtype=torch.float64, generator=g)
c = torch.zeros(nu, nu, device="cuda", dtype=torch.float64)
for name, var in (("NT", GEMM_NT), ("NN", GEMM_NN), ("TN", GEMM_TN), ("TT", GEMM_TT)):
    for kw, lab in (({}, "uniform"), ({"klo": 1}, "klo=1"), ({"khi": 1}, "khi=1"), ({"khi": 2}, "khi=2"), ({"klo": 2}, "klo=2")):
        ops.gemm_raw(var, nu, nu, nu, 1.0, a, b, 0.0, c, **kw); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gemm_raw(var, nu, nu, nu, 1.0, a, b, 0.0, c, **kw); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        fl = 2.0 * nu ** 3 * (1.0 if not kw else 0.5 + 0.5 / 64)
        print("%s %-8s %.3f ms  %.1f TFLOP/s" % (name, lab, min(ts), fl / min(ts) / 1e9), flush=True)
