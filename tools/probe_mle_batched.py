"""MLE.loss_and_grad on a batched model (params [nc, nhp]: the reference's loss.py:92-128 on x [nc, n, d]) -- K evaluations after a warm-up,
for `rocprofv3 --kernel-trace --stats` (launches per evaluation = calls / (K + 1)) and for timing.  python tools/probe_mle_batched.py nc n d [K]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygpr_amd as pg
nc, n, d = (int(a) for a in sys.argv[1:4])
K = int(sys.argv[4]) if len(sys.argv) > 4 else 4
rng = np.random.default_rng(3)
x = rng.random((nc, n, d)); y = np.sin(-x.sum(-1)) + 0.1 * rng.standard_normal((nc, n))
gp = pg.Exact_GP(torch.from_numpy(x), torch.from_numpy(y), pg.Compose([pg.Squared_exponential(), pg.White_noise()]))
hp = np.tile(np.concatenate([[1.0], np.full(d, 0.5), [0.1]]), (nc, 1))
mle = pg.MLE(gp); mle.memoize = False
mle.loss_and_grad(hp.copy()); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K): l, g = mle.loss_and_grad(hp.copy())
torch.cuda.synchronize()
print("nc=%d n=%d d=%d: %.3f ms per batched evaluation (batched path: %s), %d evaluations incl. the warm-up" % (nc, n, d, 1e3 * (time.perf_counter() - t0) / K, mle.last_batched, K + 1))
