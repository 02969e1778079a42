"""Three NLML+grad evaluations at N = 16384 with look-ahead off (single stream: the schedule bench.py's roofline leg
times launch by launch) for `rocprofv3 --kernel-trace --stats`: the profiler's GEMM-core total / calls must agree with
bench.py's roofline.avg_launch_ms."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygpr_amd as pg
import bench
from pygpr_amd._ops import get_ops
n, d = 16384, 8
x, y = bench.synth_expert(n, d, 1234)
gp = pg.Exact_GP(torch.from_numpy(x), torch.from_numpy(y), pg.Compose([pg.Squared_exponential(), pg.White_noise()]))
mle = pg.MLE(gp)
mle.memoize = False
hp = np.concatenate([[1.0], np.ones(d), [0.1]])
get_ops().set_lookahead(0)
for i in range(3):
    l, g = mle.loss_and_grad(hp * (1 + 1e-3 * i))
torch.cuda.synchronize()
print("loss", float(l))
