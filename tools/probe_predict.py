"""Config-2 predict leg and config-4 committee throughput (the launches that use the column-sum-of-squares epilogue)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygpr_amd as pg
import bench
d = 8
cov = pg.Compose([pg.Squared_exponential(), pg.White_noise()])
x2, y2 = bench.synth_expert(8192, d, 4242)
gp2 = pg.Exact_GP(torch.from_numpy(x2), torch.from_numpy(y2), cov, eager_inverse=True)
gp2.set_params(torch.from_numpy(bench.default_hp(d)))
xs2 = torch.from_numpy(np.random.default_rng(4321).random((8192, d))).cuda()
gp2.update(); gp2.predict(xs2, var="diag"); torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t1 = time.perf_counter(); gp2.predict(xs2, var="diag"); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t1)
print("cfg2 predict %.3f ms" % (best * 1e3))
