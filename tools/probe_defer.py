"""Deferred trailing block of the coupled-chain factorisation (linalg.hip, round 5): time and correctness of pg_potrf and of the fused
pg_build_potrf_trtri at the given sizes under the environment's PG_DEFER* settings (read once per process: run one process per setting).

    python tools/probe_defer.py 8192 4096 6144            # build + potrf, potrf alone (build time subtracted), fused, residual vs LAPACK
    PG_DEFER=0 python tools/probe_defer.py 8192

Prints one line per size: `n potrf_ms fused_ms deferred_panels coupled_panels max|L - L_lapack|`.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops, make_spec  # noqa: E402

ops = get_ops()
check = os.environ.get("PROBE_CHECK", "1") != "0"


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


for n in ([int(a) for a in sys.argv[1:]] or [8192]):
    d = 8
    rng = np.random.default_rng(1234)
    xh = rng.random((n, d))
    x = torch.from_numpy(xh).cuda()
    hp = torch.tensor([1.0] + [1.0] * d + [0.1], dtype=torch.float64).cuda()
    spec = make_spec([0], [0], [d + 1])
    kl = ops.empty(n, n)
    invd = ops.potrf_workspace(n, torch.float64)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    minv = ops.empty(n, n)

    def build():
        ops.kernel_build(spec, hp, x, None, kl, lower_only=True, jitter=1e-7)

    def fac():
        build()
        ops.potrf(kl, invd, info)

    def fused():
        ops.build_factor(spec, hp, x, kl, invd, info, minv=minv, jitter=1e-7)

    tb = timed(build, 3)
    tf = timed(fac) - tb
    dp, cp = ops.last_deferred_panels(), ops.last_coupled_panels()
    tfu = timed(fused)
    err = float("nan")
    if check:
        fac()
        torch.cuda.synchronize()
        L = torch.tril(kl).cpu().numpy()
        d2 = ((xh[:, None, :] - xh[None, :, :]) ** 2).sum(-1) if n <= 2048 else None
        if d2 is None:
            sq = (xh ** 2).sum(1)
            d2 = np.maximum(sq[:, None] + sq[None, :] - 2 * xh @ xh.T, 0)
        K = np.exp(-d2) + (0.01 + 1e-7) * np.eye(n)
        Lr = np.linalg.cholesky(K)
        err = float(np.abs(L - Lr).max())
        fused()
        torch.cuda.synchronize()
        Mi = torch.tril(minv).cpu().numpy()
        r = np.abs(Mi[: min(n, 2048)] @ Lr - np.eye(n)[: min(n, 2048)]).max()
        err = max(err, float(r))
    print(f"n={n} potrf {tf:.3f} ms ({n**3/3/tf/1e9:.1f} TF/s, {n**3/3/tf/1e9/78.6:.3f})  build+fused {tfu:.3f} ms  deferred={dp} coupled={cp} "
          f"info={int(info.item())} err={err:.2e}", flush=True)
    del kl, minv, invd
