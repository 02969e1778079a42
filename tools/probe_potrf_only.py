"""potrf alone (covariance build subtracted) at the given sizes; env knobs (PG_SYNC_ROWS, PG_CS_PANEL, PG_NBO ...) are read by the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pygpr_amd._ops import get_ops, make_spec
ops = get_ops()
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
out = []
for n in [int(a) for a in sys.argv[1:]] or [4096, 8192]:
    d = 8
    x = torch.from_numpy(np.random.default_rng(1234).random((n, d))).cuda()
    hp = torch.tensor([1.0] + [1.0] * d + [0.1], dtype=torch.float64).cuda()
    spec = make_spec([0], [0], [d + 1])
    kl = ops.empty(n, n); invd = ops.potrf_workspace(n, torch.float64); info = torch.zeros(1, dtype=torch.int32, device="cuda")
    tb = ev(lambda: ops.kernel_build(spec, hp, x, None, kl, lower_only=True, jitter=1e-7))
    def run():
        ops.kernel_build(spec, hp, x, None, kl, lower_only=True, jitter=1e-7); ops.potrf(kl, invd, info)
    t = ev(run) - tb
    out.append("n=%d potrf %.3f ms (%.1f TF/s, %.3f of peak) coupled panels %d info %d" % (n, t, n ** 3 / 3 / t / 1e9, n ** 3 / 3 / t / 1e9 / 78.6, ops.last_coupled_panels(), int(info.item())))
    del kl, invd
print(" | ".join(out), flush=True)
