"""Exact_GP.predict(var="diag") on a batched model (the reference's gpr.py:76-106 on x [nc, n, d]) -- K predictions after a warm-up, for
`rocprofv3 --kernel-trace --stats` (launches per prediction = calls / (K + 1) once the fit's kernels are subtracted: run with K = 0 for those)
and for timing.  PG_PREDICT_SERIAL=1 in the environment walks the experts one by one (rounds 1-4).
python tools/probe_predict_batched.py nc n d m [K]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygpr_amd as pg
nc, n, d, m = (int(a) for a in sys.argv[1:5])
K = int(sys.argv[5]) if len(sys.argv) > 5 else 4
rng = np.random.default_rng(3)
x = rng.random((nc, n, d)); y = np.sin(-x.sum(-1)) + 0.1 * rng.standard_normal((nc, n))
gp = pg.Exact_GP(torch.from_numpy(x), torch.from_numpy(y), pg.Compose([pg.Squared_exponential(), pg.White_noise()]), eager_inverse=True)
gp.set_params(torch.from_numpy(np.tile(np.concatenate([[1.0], np.full(d, 0.5), [0.1]]), (nc, 1))))
xs = torch.from_numpy(rng.random((m, d))).cuda()
gp.update(); torch.cuda.synchronize()
if K == 0:
    print("fit only (nc=%d n=%d d=%d): subtract this run's kernel counts from a K > 0 run" % (nc, n, d)); sys.exit(0)
gp.predict(xs, var="diag"); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K): mu, var = gp.predict(xs, var="diag")
torch.cuda.synchronize()
print("nc=%d n=%d d=%d m=%d: %.3f ms per prediction (batched path: %s), %d predictions incl. the warm-up" % (nc, n, d, m, 1e3 * (time.perf_counter() - t0) / K, gp.last_predict_batched, K + 1))
