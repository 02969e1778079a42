import sys, numpy as np, torch
sys.path.insert(0, ".")
from pygpr_amd._ops import get_ops, make_spec
ops = get_ops()
n = 16384
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
k = ops.empty(n, n)
for d in (8, 2, 16):
    rng = np.random.default_rng(d)
    x = torch.from_numpy(rng.random((n, d))).cuda()
    hp = torch.tensor([1.0] + [1.0] * d + [0.1], dtype=torch.float64).cuda()
    for kind, name in ((0, "rbf"),):
        spec = make_spec([kind], [0], [d + 1])
        tl = ev(lambda: ops.kernel_build(spec, hp, x, None, k, lower_only=True, jitter=1e-7))
        tf = ev(lambda: ops.kernel_build(spec, hp, x, None, k, jitter=1e-7))
        print(f"d={d:2d} {name:6s} lower {tl:.3f} ms {(4*n*(n+64)+8*n*d)/tl/1e6:.0f} GB/s   full(mirror) {tf:.3f} ms  {8*n*n/tf/1e6:.0f} GB/s", flush=True)
