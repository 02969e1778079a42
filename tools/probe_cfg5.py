"""BASELINE config 5 regime on one GPU: one expert, n = 1024 + 32768, D = 8, Matern-5/2 + noise, fp32."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import pygpr_amd as pg
n, d = 33792, 8
rng = np.random.default_rng(5)
x = rng.random((n, d)); y = np.sin(-x.sum(1)) + 0.1 * rng.standard_normal(n)
cov = pg.Compose([pg.Matern52(), pg.White_noise()])
hp = np.concatenate([[1.0], np.ones(d), [0.1]])
for dt in (torch.float32, torch.float64):
    gp = pg.Exact_GP(torch.from_numpy(x).to(dt), torch.from_numpy(y).to(dt), cov)
    mle = pg.MLE(gp)
    mle.memoize = False
    l, g = mle.loss_and_grad(hp.copy()); torch.cuda.synchronize()
    t = time.perf_counter(); l, g = mle.loss_and_grad(hp.copy()); torch.cuda.synchronize(); t = time.perf_counter() - t
    print(f"{dt}: n={n} loss={float(l):.6f} |g|inf={np.abs(g).max():.4f} g0={g[0]:.4f} eval {t*1e3:.1f} ms  {n**3/t/1e12:.1f} TFLOP/s eff", flush=True)
    del gp, mle; torch.cuda.empty_cache()
