"""A loop of NLML+gradient evaluations at N = 16384 exactly as bench.py's timed loop issues them (for kernel traces)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygpr_amd as pg
import bench
n, d = 16384, 8
x, y = bench.synth_expert(n, d, 1234)
gp = pg.Exact_GP(torch.from_numpy(x), torch.from_numpy(y), pg.Compose([pg.Squared_exponential(), pg.White_noise()]))
mle = pg.MLE(gp); mle.memoize = False
hp = np.concatenate([[1.0], np.ones(d), [0.1]])
for i in range(2): mle.loss_and_grad(hp)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for i in range(K): l, g = mle.loss_and_grad(hp)
torch.cuda.synchronize(); print("ms per evaluation %.3f" % (1e3 * (time.perf_counter() - t0) / K), float(l))
