"""One batched Exact_GP.update() (eager inverse) of nc experts of n points, repeated, for kernel traces."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygpr_amd as pg
nc, n, d = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 4096, 16)))
rng = np.random.default_rng(3)
x = rng.random((nc, n, d)); y = np.sin(-x.sum(-1)) + 0.1 * rng.standard_normal((nc, n))
gp = pg.Exact_GP(torch.from_numpy(x), torch.from_numpy(y), pg.Compose([pg.Squared_exponential(), pg.White_noise()]), eager_inverse=True)
gp.set_params(torch.from_numpy(np.tile(np.concatenate([[1.0], np.full(d, 0.5), [0.1]]), (nc, 1))))
gp.update(); torch.cuda.synchronize()
best = 1e9
for _ in range(4):
    gp.need_upd = True; t0 = time.perf_counter(); gp.update(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print("fit %d x %d: %.3f ms" % (nc, n, 1e3 * best))
