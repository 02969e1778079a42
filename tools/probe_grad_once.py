"""Two launches of the gradient contraction alone (N = 16384, fp64, one squared-exponential component; D from argv, default 16) for
counter passes: rocprofv3 --pmc <counters> -- python3 tools/probe_grad_once.py [D]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops, make_spec
ops = get_ops()
n, d = 16384, (int(sys.argv[1]) if len(sys.argv) > 1 else 16)
rng = np.random.default_rng(1)
x = torch.from_numpy(rng.random((n, d))).cuda()
hp = torch.tensor([1.0] + [0.5] * d + [0.1], dtype=torch.float64).cuda()
spec = make_spec([0], [0], [d + 1])
g = torch.Generator(device="cuda").manual_seed(3)
kinv = torch.randn(n, n, device="cuda", dtype=torch.float64, generator=g)
alpha = torch.randn(n, device="cuda", dtype=torch.float64, generator=g)
grad = ops.zeros(d + 2); work = ops.empty(ops.nlml_grad_worksize(n, d + 2))
for _ in range(2):
    ops.nlml_grad(spec, hp, x, n, kinv, alpha, grad, work)
torch.cuda.synchronize()
print("grad[:3]", grad.cpu().numpy()[:3])
