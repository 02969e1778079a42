"""One look-ahead potrf at n = 16384 for a kernel-trace timeline (run under rocprofv3 --kernel-trace)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops, make_spec
ops = get_ops()
n, d = (int(sys.argv[2]) if len(sys.argv) > 2 else 16384), 8
rng = np.random.default_rng(1234)
x = torch.from_numpy(rng.random((n, d))).cuda()
hp = torch.tensor([1.0] + [1.0] * d + [0.1], dtype=torch.float64).cuda()
spec = make_spec([0], [0], [d + 1])
kl = ops.empty(n, n)
FUSED = len(sys.argv) > 1 and sys.argv[1] == "fused"
minv = ops.empty(n, n) if FUSED else None
invd = ops.potrf_workspace(n, torch.float64); info = torch.zeros(1, dtype=torch.int32, device="cuda")
for _ in range(2):
    ops.kernel_build(spec, hp, x, None, kl, lower_only=True, jitter=1e-7)
    torch.cuda.synchronize()
    (ops.potrf_trtri(kl, invd, info, minv) if FUSED else ops.potrf(kl, invd, info))
    torch.cuda.synchronize()
print("info", int(info.item()))
