import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpr_amd._ops import get_ops
ops = get_ops()
rng = np.random.default_rng(0)
nb = 64
mats = []
for _ in range(nb):
    a = rng.standard_normal((128, 128)); mats.append(a @ a.T / 128 + np.eye(128))
src = torch.from_numpy(np.stack(mats)).cuda()
inv = torch.zeros(nb, 128, 128, device="cuda", dtype=torch.float64)
info = torch.zeros(1, dtype=torch.int32, device="cuda")
def run(ablate, with_inv=True):
    a = src.clone(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for b in range(nb): ops.leaf_raw(a[b], inv[b] if with_inv else None, info, ablate)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nb * 1e3
for name, ab, wi in [("full", 0, True), ("full, progressive loads/stores", 16, True), ("full, square-root-free chain", 32, True), ("full, both", 48, True),  ("no inverse", 2, True), ("no factor loop (I/O + tail of the inverse)", 1, True),
                     ("no factor loop, no inverse (I/O + launch only)", 3, True), ("no diagonal step (A)", 8, True),
                     ("no A, no inverse", 10, True)]:
    run(ab, wi); t = min(run(ab, wi) for _ in range(3))
    print(f"{name:50s} {t:7.1f} us per leaf", flush=True)
