"""Exact_GP at config 2's size: default (lazy) update vs eager_inverse, mean-only and mean+variance prediction."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygpr_amd as pg
import bench
n, d, m = 8192, 8, 8192
x, y = bench.synth_expert(n, d, 1234)
xs = torch.from_numpy(np.random.default_rng(4321).random((m, d)))
cov = pg.Compose([pg.Squared_exponential(), pg.White_noise()])
hp = torch.from_numpy(np.concatenate([[1.0], np.ones(d), [0.1]]))
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
for eager in (False, True):
    gp = pg.Exact_GP(torch.from_numpy(x), torch.from_numpy(y), cov, eager_inverse=eager)
    def fit():
        gp.set_params(hp); gp.update()
    print(f"eager_inverse={eager}: update {t(fit):.2f} ms", end="")
    print(f"  predict mean only {t(lambda: gp.predict(xs, var=None)):.2f} ms", end="")
    def fit_var():
        gp.set_params(hp); gp.update(); gp.predict(xs, var='diag')
    print(f"  update + predict(diag) {t(fit_var):.2f} ms   predict(diag) again {t(lambda: gp.predict(xs, var='diag')):.2f} ms", flush=True)
