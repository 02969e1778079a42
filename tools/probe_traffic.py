"""Kernels whose HBM traffic we want from PMC counters: covariance build (full, n=16384) and one long-K lower SYRK."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from pygpr_amd._ops import get_ops, make_spec
from pygpr_amd._lib import GEMM_NT
ops = get_ops()
n, d = 16384, 8
rng = np.random.default_rng(1234)
x = torch.from_numpy(rng.random((n, d))).cuda()
hp = torch.tensor([1.0] + [1.0] * d + [0.1], dtype=torch.float64).cuda()
spec = make_spec([0], [0], [d + 1])
k = ops.empty(n, n)
for _ in range(3):
    ops.kernel_build(spec, hp, x, None, k, jitter=1e-7)
torch.cuda.synchronize()
p = k[:, :1024].contiguous()
c = ops.zeros(n, n)
for _ in range(3):
    ops.gemm_raw(GEMM_NT, n, n, 1024, -1.0, p, p, 1.0, c, tri=1)
torch.cuda.synchronize()
