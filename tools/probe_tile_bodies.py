"""The covariance build (lower-only, mirrored, cross) and the gradient contraction on their own, matrix-pipe bodies (kmfma.hip) against
the VALU bodies (PG_KB_MFMA=0 / PG_GRAD_MFMA=0), by HIP events: fp64 squared exponential at D = 8 / 16 (N = 16384) and fp32 Matern-5/2 at
D = 16 (n = 33792: BASELINE config 5's expert).  python tools/probe_tile_bodies.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops, make_spec
ops = get_ops()
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(4): fn()
        b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b) / 4)
    return best
for n, d, kind, dt in ((16384, 8, 0, torch.float64), (16384, 16, 0, torch.float64), (16384, 16, 1, torch.float64), (33792, 16, 1, torch.float32), (16384, 8, 0, torch.float32)):
    item = 8 if dt == torch.float64 else 4
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.random((n, d))).cuda().to(dt)
    hp = torch.tensor([1.0] + [0.5] * d + [0.1], dtype=torch.float64).cuda()
    spec = make_spec([kind], [0], [d + 1])
    g = torch.Generator(device="cuda").manual_seed(3)
    kinv = torch.randn(n, n, device="cuda", dtype=dt, generator=g)
    alpha = torch.randn(n, device="cuda", dtype=dt, generator=g)
    grad = ops.zeros(d + 2); work = ops.empty(ops.nlml_grad_worksize(n, d + 2))
    k = ops.empty(n, n, dtype=dt)
    m = 8192
    xs = x[:m].contiguous(); kx = ops.empty(m, n, dtype=dt)
    res = {}
    for mode in ("mfma", "valu"):
        os.environ["PG_KB_MFMA"] = "2" if mode == "mfma" else "0"
        os.environ["PG_GRAD_MFMA"] = "1" if mode == "mfma" else "0"
        tg = ev(lambda: ops.nlml_grad(spec, hp, x, n, kinv, alpha, grad, work))
        gv = grad.cpu().numpy().copy()
        tl = ev(lambda: ops.kernel_build(spec, hp, x, None, k, lower_only=True, jitter=1e-7))
        tf = ev(lambda: ops.kernel_build(spec, hp, x, None, k, jitter=1e-7))
        tx = ev(lambda: ops.kernel_build(spec, hp, xs, x, kx))
        res[mode] = gv
        print(f"n={n} d={d} kind={'rbf' if kind == 0 else 'matern52'} {str(dt)[6:]} {mode}: grad {tg*1e3:.0f} us ({item/2*n*n/tg/1e6:.0f} GB/s of K^-1)  "
              f"build lower {tl*1e3:.0f} us ({item/2*n*(n+64)/tl/1e6:.0f} GB/s)  mirrored {tf*1e3:.0f} us ({item*n*n/tf/1e6:.0f} GB/s)  "
              f"cross {m}x{n} {tx*1e3:.0f} us ({item*m*n/tx/1e6:.0f} GB/s)", flush=True)
    print("   gradient, matrix pipe vs VALU: max rel diff %.2e" % (np.abs(res["mfma"] - res["valu"]).max() / np.abs(res["valu"]).max()), flush=True)
    del kinv, k, kx
    torch.cuda.empty_cache()
