"""The fused gradient contraction (pg_nlml_grad) and the lower-only covariance build on their own at N = 16384."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops, make_spec
ops = get_ops()
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
n = 16384
for d in (8, 16):
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.random((n, d))).cuda()
    hp = torch.tensor([1.0] + [0.7] * d + [0.1], dtype=torch.float64).cuda()
    spec = make_spec([0], [0], [d + 1])
    g = torch.Generator(device="cuda").manual_seed(3)
    kinv = torch.randn(n, n, device="cuda", dtype=torch.float64, generator=g)
    alpha = torch.randn(n, device="cuda", dtype=torch.float64, generator=g)
    grad = ops.zeros(d + 2); work = ops.empty(ops.nlml_grad_worksize(n, d + 2))
    t = ev(lambda: ops.nlml_grad(spec, hp, x, n, kinv, alpha, grad, work))
    print("grad", grad.cpu().numpy())
    k = ops.empty(n, n)
    tb = ev(lambda: ops.kernel_build(spec, hp, x, None, k, lower_only=True, jitter=1e-7))
    tf = ev(lambda: ops.kernel_build(spec, hp, x, None, k, jitter=1e-7))
    print(f"d={d}: nlml_grad {t*1e3:.0f} us; kernel build lower-only {tb*1e3:.0f} us ({4.0*n*(n+64)/tb/1e6:.0f} GB/s), full {tf*1e3:.0f} us ({8.0*n*n/tf/1e6:.0f} GB/s)", flush=True)
    del kinv, k
