"""BASELINE config 5 as a committee on ONE GPU: grBCM, 8 local experts x (ng 1024 + nls 32768), D = 16,
Matern-5/2 + noise, fp32, shared hyper-parameters; 10 evaluations of the co-training objective sum_c NLML_c
and its gradient (GRBCM_MLE).  With one expert per GPU (the 8-GPU layout) the per-rank time is 1/8 of this
plus one all-reduce of [1 + nhp] doubles."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygpr_amd as pg

nc, nls, ng, d = 8, 32768, 1024, 16
evals = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(1234)
f = lambda x: np.sin(-x.sum(-1)) + 0.1 * rng.standard_normal(x.shape[:-1])
xl = rng.random((nc, nls, d)); yl = f(xl)
xg = rng.random((ng, d)); yg = f(xg)
t32 = lambda a: torch.from_numpy(a).to(torch.float32)
cov = pg.Compose([pg.Matern52(), pg.White_noise()])
model = pg.GRBCM(t32(xl), t32(yl), t32(xg), t32(yg), cov)
obj = pg.GRBCM_MLE(model)
obj.memoize = False
hp = np.concatenate([[1.0], 0.5 * np.ones(d), [0.1]])
l, g = obj.loss_and_grad(hp.copy()); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(evals):
    l, g = obj.loss_and_grad(hp * (1.0 + 1e-3 * (i + 1)))
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / evals
n = ng + nls
flop = nc * float(n) ** 3 + float(ng) ** 3
print(f"cfg5 co-training, 1 GPU: {nc} experts x n={n}, D={d}, Matern-5/2, fp32: {t*1e3:.1f} ms per evaluation "
      f"({1/t:.3f} evals/s, {t/nc*1e3:.1f} ms per expert, {flop/t/1e12:.1f} TFLOP/s effective = "
      f"{flop/t/1e12/157.3*100:.0f} % of the fp32 matrix peak); loss {float(l):.4f} |g|inf {np.abs(g).max():.4f}", flush=True)
