"""GRBCM.predict(var='full') at config 4's size (8 experts x 9216 points, m = 2048), three calls after the fit and a warm-up: the
profiling input for the kernel statistics of the bench leg `grbcm_predict_full`."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygpr_amd as pg
nc, nl, ng, d, m = 8, 8192, 1024, 16, 2048
def synth(n, seed):
    rng = np.random.default_rng(seed); x = rng.random((n, d)); return x, np.sin(-x.sum(1)) + 0.1 * rng.standard_normal(n)
xg, yg = synth(ng, 7); sh = [synth(nl, 100 + c) for c in range(nc)]
g = pg.GRBCM(torch.from_numpy(np.stack([s[0] for s in sh])), torch.from_numpy(np.stack([s[1] for s in sh])), torch.from_numpy(xg), torch.from_numpy(yg),
             pg.Compose([pg.Squared_exponential(), pg.White_noise()]))
hp = torch.from_numpy(np.concatenate([[1.0], np.full(d, 0.5), [0.1]])); g.gpg.set_params(hp); g.set_local_params(hp)
xs = torch.from_numpy(np.random.default_rng(4321).random((m, d))).cuda()
g.predict(xs, var="full"); torch.cuda.synchronize()
torch.cuda.nvtx.range_push("three calls") if hasattr(torch.cuda, "nvtx") else None
for _ in range(3): mu, cov = g.predict(xs, var="full")
torch.cuda.synchronize(); print("ok", float(cov.diagonal().mean()))
