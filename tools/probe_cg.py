"""BASELINE config 3 end to end: scipy CG (PyGPR opt.py:45-67) with maxiter=50 on N=16384, D=8, plus 50 raw evaluations."""
import sys, time, os, tempfile, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(tempfile.mkdtemp())
import pygpr_amd as pg
n, d = 16384, 8
rng = np.random.default_rng(1234)
x = rng.random((n, d)); y = np.sin(-x.sum(1)) + 0.1 * rng.standard_normal(n)
gp = pg.Exact_GP(torch.from_numpy(x), torch.from_numpy(y), pg.Compose([pg.Squared_exponential(), pg.White_noise()]))
hp0 = np.concatenate([[1.0], np.ones(d), [0.1]])
gp.set_params(torch.from_numpy(hp0))
mle = pg.MLE(gp)
mle.memoize = False
mle.loss_and_grad(hp0); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(50): l, g = mle.loss_and_grad(hp0)
torch.cuda.synchronize(); t = time.perf_counter() - t
print(f"50 raw evaluations: {t:.2f} s  -> {50/t:.2f} evals/s; NLML {float(l):.4f}", flush=True)
cg = pg.CG(mle); cg.args.update(maxiter=50, disp=False)
t = time.perf_counter(); cg.minimize(); torch.cuda.synchronize(); t = time.perf_counter() - t
print(f"CG maxiter=50: {t:.2f} s, nit={cg.res.nit} nfev={cg.res.nfev} ({cg.res.nfev/t:.2f} evals/s), NLML {hp0 is not None and float(l):.2f} -> {float(cg.res.fun):.4f}, success={cg.res.success}")
print("final hp", np.array2string(cg.res.x, precision=4))
