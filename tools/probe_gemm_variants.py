"""Uniform 8192^3 products in the four operand layouts of the GEMM core (for counter passes: LDS bank conflicts per layout)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops
from pygpr_amd._lib import GEMM_NT, GEMM_NN, GEMM_TN, GEMM_TT
ops = get_ops()
nu = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn(nu, nu, device="cuda", dtype=torch.float64, generator=g)
b = torch.randn(nu, nu, device="cuda", dtype=torch.float64, generator=g)
c = torch.zeros(nu, nu, device="cuda", dtype=torch.float64)
for name, var in (("NT", GEMM_NT), ("NN", GEMM_NN), ("TN", GEMM_TN), ("TT", GEMM_TT)):
    ops.gemm_raw(var, nu, nu, nu, 1.0, a, b, 0.0, c)
    torch.cuda.synchronize()
print("done")
