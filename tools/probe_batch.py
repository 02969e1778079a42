"""Experts together: time Exact_GP.update() (covariance build + Cholesky + L^-1 + alpha per expert) for nc experts of n points,
batched call vs one expert after the other.  PG_BATCH_MAX_N=0 forces the serial loop.  python tools/probe_batch.py nc n [d]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pygpr_amd as pg
from pygpr_amd._ops import get_ops
nc, n = int(sys.argv[1]), int(sys.argv[2])
d = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rng = np.random.default_rng(3)
x = rng.random((nc, n, d)); y = np.sin(-x.sum(-1)) + 0.1 * rng.standard_normal((nc, n))
hp = np.concatenate([[1.0], np.full(d, 0.5), [0.1]])
for eager in (True, False):
    gp = pg.Exact_GP(torch.from_numpy(x), torch.from_numpy(y), pg.Compose([pg.Squared_exponential(), pg.White_noise()]), eager_inverse=eager)
    gp.set_params(torch.from_numpy(np.tile(hp, (nc, 1))))
    gp.update(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        gp.need_upd = True
        t0 = time.perf_counter(); gp.update(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    flop = nc * (n ** 3 / 3.0) * (2 if eager else 1)
    print("PG_BATCH_MAX_N=%s nc=%d n=%d eager_inverse=%s: update %.2f ms (%.1f TFLOP/s) batched=%s coupled panels %d" % (
        os.environ.get("PG_BATCH_MAX_N", "default"), nc, n, eager, 1e3 * best, flop / best / 1e12, gp._bat is not None, get_ops().last_coupled_panels()), flush=True)
    del gp; torch.cuda.empty_cache()
