"""potrf / trtri / lauum only, for rocprofv3 --kernel-trace --stats."""
import sys
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops, make_spec
ops = get_ops()
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 8
rng = np.random.default_rng(1234)
x = torch.from_numpy(rng.random((n, d))).cuda()
hp = torch.tensor([1.0] + [1.0] * d + [0.1], dtype=torch.float64).cuda()
spec = make_spec([0], [0], [d + 1])
k = ops.empty(n, n); kl = ops.empty(n, n)
ops.kernel_build(spec, hp, x, None, k, jitter=1e-7)
invd = ops.potrf_workspace(n, torch.float64); info = torch.zeros(1, dtype=torch.int32, device="cuda")
minv = ops.zeros(n, n); kinv = ops.zeros(n, n)
for _ in range(3):
    kl.copy_(k); ops.potrf(kl, invd, info)
    if len(sys.argv) < 3: ops.trtri(kl, invd, minv); ops.lauum(minv, kinv)
torch.cuda.synchronize()
print("info", int(info.item()))
