"""One NLML+grad evaluation at N = 16384 (after one warm-up) for counter passes:
  rocprofv3 --pmc FETCH_SIZE -d out_f --output-format csv -- python3 tools/probe_eval_once.py
  rocprofv3 --pmc WRITE_SIZE -d out_w --output-format csv -- python3 tools/probe_eval_once.py
then  python tools/pmc_summary.py out_f out_w > profiles/<round>_pmc_eval_traffic.json"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygpr_amd as pg
import bench
n, d = 16384, 8
x, y = bench.synth_expert(n, d, 1234)
gp = pg.Exact_GP(torch.from_numpy(x), torch.from_numpy(y), pg.Compose([pg.Squared_exponential(), pg.White_noise()]))
mle = pg.MLE(gp)
mle.memoize = False
hp = np.concatenate([[1.0], np.ones(d), [0.1]])
for i in range(2):
    l, g = mle.loss_and_grad(hp * (1 + 1e-3 * i))
torch.cuda.synchronize()
print("loss", float(l), "coupled chain enabled:", pg._ops.get_ops().coupled_chain(), "panels coupled in the last call:", pg._ops.get_ops().last_coupled_panels())
