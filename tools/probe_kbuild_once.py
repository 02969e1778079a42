"""Covariance build alone for rocprofv3: 3 lower-only then 3 mirrored symmetric builds at N = 16384, D = 8, fp64 (Compose([SE, WN]))."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pygpr_amd._ops import get_ops, make_spec
ops = get_ops()
n, d = 16384, 8
x = torch.from_numpy(np.random.default_rng(8).random((n, d))).cuda()
hp = torch.tensor([1.0] + [1.0] * d + [0.1], dtype=torch.float64).cuda()
spec = make_spec([0], [0], [d + 1])
k = ops.empty(n, n)
for lower in (True, True, True, False, False, False):
    ops.kernel_build(spec, hp, x, None, k, lower_only=lower, jitter=1e-7)
    torch.cuda.synchronize()
print("done")
