"""NLML + gradient at a size that pads unevenly across the recursive split (n = 16500 -> 16640 = 8448 + 8192, 140 rows of identity padding in
the trailing half): run once with PG_REC_MIN=0 and once with the default, the two lines must agree to rounding."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygpr_amd as pg
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 16500, 4
rng = np.random.default_rng(5)
x = rng.random((n, d)); y = np.sin(3 * x.sum(1)) + 0.1 * rng.standard_normal(n)
hp = np.concatenate([[1.1], np.full(d, 0.8), [0.2]])
mle = pg.MLE(pg.Exact_GP(torch.from_numpy(x), torch.from_numpy(y), pg.Compose([pg.Squared_exponential(), pg.White_noise()])))
l, g = mle.loss_and_grad(hp.copy())
print("REC_MIN=%s n=%d nlml %.12f grad %s" % (os.environ.get("PG_REC_MIN", "default"), n, float(l), np.array2string(g, precision=10)))
