"""Accuracy of the covariance build against a long-double evaluation of sig^2 exp(-sum((x - x') l)^2) + diag: max relative error of K
(symmetric mirrored build, n = 2048, d = 8 and 16; coordinates in [0, 1) and in [0, 30)).  PG_KB_FAST=0 / 1 selects the body."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pygpr_amd._ops import get_ops, make_spec
ops = get_ops()
n = 2048
for d, span in ((8, 1.0), (16, 1.0), (8, 30.0)):
    rng = np.random.default_rng(d)
    xh = rng.random((n, d)) * span
    l = rng.uniform(0.3, 1.2, d) / np.sqrt(span)
    hph = np.concatenate([[1.3], l, [0.1]])
    x = torch.from_numpy(xh).cuda(); hp = torch.from_numpy(hph).cuda()
    spec = make_spec([0], [0], [d + 1])
    k = ops.empty(n, n)
    ops.kernel_build(spec, hp, x, None, k, jitter=0.0)
    torch.cuda.synchronize()
    kh = k.cpu().numpy()[:n, :n]
    xs = (xh * l).astype(np.longdouble)
    sq = ((xs[:, None, :] - xs[None, :, :]) ** 2).sum(-1)
    ref = (np.longdouble(1.3) ** 2) * np.exp(-sq) + np.eye(n, dtype=np.longdouble) * np.longdouble(0.1) ** 2
    rel = np.abs(kh - ref) / np.abs(ref)
    print("PG_KB_FAST=%s d=%d span=%g: max rel err %.3e  (mean %.3e), symmetric: %s, min K %.3e" % (
        os.environ.get("PG_KB_FAST", "default"), d, span, float(rel.max()), float(rel.mean()), bool((kh == kh.T).all()), float(kh.min())), flush=True)
