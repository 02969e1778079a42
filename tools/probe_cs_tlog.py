"""Timestamps inside the flag-coupled chain (PG_CS_TLOG=1): per 128-column step, from the leaf and the first rows workgroup.
python tools/probe_cs_tlog.py [n]   (n <= 8192 so that every panel is coupled)"""
import os, sys
os.environ["PG_CS_TLOG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pygpr_amd._ops import get_ops, make_spec
ops = get_ops()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = 8
rng = np.random.default_rng(1234)
x = torch.from_numpy(rng.random((n, d))).cuda()
hp = torch.tensor([1.0] + [1.0] * d + [0.1], dtype=torch.float64).cuda()
spec = make_spec([0], [0], [d + 1])
kl = ops.empty(n, n)
invd = ops.potrf_workspace(n, torch.float64); info = torch.zeros(1, dtype=torch.int32, device="cuda")
for _ in range(3):
    invd.zero_()
    ops.kernel_build(spec, hp, x, None, kl, lower_only=True, jitter=1e-7); ops.potrf(kl, invd, info)
    torch.cuda.synchronize()
nblk = n // 128
base = n * 128                              # doubles: start of the flag words (right behind the diagonal-block inverses)
tmo_off_bytes = base * 8 + 5 * nblk * 4
tl = invd.view(torch.int64)[(tmo_off_bytes // 8) + 512: (tmo_off_bytes // 8) + 512 + 48 * nblk].cpu().numpy().reshape(nblk, 48)
# wall_clock64: 100 MHz
t0 = tl[0, 0]
us = lambda v: (v - t0) / 100.0
print("step | leaf: start flag-seen body-end drained | rows wg0: before-wait flag-seen T-done published brow-seen U-done published  (us; deltas)")
for k in range(1, min(nblk - 1, 14)):
    L = tl[k]; Lp = tl[k - 1]
    print("%3d leaf start %8.1f | wait %5.1f body %5.1f drain %4.1f || rows: done-flag seen +%4.1f after leaf drained; T %4.1f; publish %4.1f; brow wait %4.1f; U %4.1f; publish %4.1f | next leaf sees tile +%4.1f" % (
        k, us(L[0]), (L[1] - L[0]) / 100, (L[2] - L[1]) / 100, (L[3] - L[2]) / 100,
        (L[5] - L[3]) / 100, (L[6] - L[5]) / 100, (L[7] - L[6]) / 100, (L[8] - L[7]) / 100, (L[9] - L[8]) / 100, (L[10] - L[9]) / 100,
        (tl[k + 1][1] - L[10]) / 100))

print("leaf phases (us): load | per micro-panel: step (F: the diagonal block's factor alone, wave 0), first update column | tail to body-end")
for k in range(1, min(nblk - 1, 10)):
    L = tl[k]
    parts = ["load %.1f" % ((L[16] - L[1]) / 100)]
    prev = L[16]
    for jb in range(8):
        a = L[17 + 2 * jb]
        b = L[18 + 2 * jb] if jb < 7 else a
        parts.append("%d: %.2f(F %.2f)+%.2f" % (jb, (a - prev) / 100, (L[35 + jb] - prev) / 100, (b - a) / 100))
        prev = b
    parts.append("finish %.2f, last stores %.2f" % ((L[34] - prev) / 100, (L[2] - L[34]) / 100))
    print("%3d " % k + " | ".join(parts))


# shader clock held during each leaf: s_memtime ticks over s_memrealtime ticks (100 MHz) around the body (MI355X_MICROARCH.md, DVFS item 6)
clk = [(tl[k][45] - tl[k][44]) / max(tl[k][2] - tl[k][1], 1) * 0.1 for k in range(1, nblk - 1) if tl[k][45] > tl[k][44]]
if clk:
    print("in-kernel shader clock during the leaves (GHz): first %.2f, median %.2f, min %.2f, max %.2f, last %.2f (%d leaves)" % (
        clk[0], sorted(clk)[len(clk) // 2], min(clk), max(clk), clk[-1], len(clk)))

# one worker wave's slot beside the factor of micro-panel 3 (wave 1, slot of step jb = 2): from the barrier behind M to its start,
# its tiles (deferred trailing update + the inverse's block row), its stores of the previous step's panel / block row
ws = [((tl[k][43] - tl[k][22]) / 100, (tl[k][46] - tl[k][43]) / 100, (tl[k][38] - tl[k][22]) / 100, (tl[k][23] - tl[k][22]) / 100) for k in range(1, nblk - 1)]
st = [(tl[k][47] - tl[k][46]) / 100 for k in range(1, nblk - 1) if tl[k][47] > tl[k][46]]     # stamp 47 is only written by leaf forms whose workers store (an unset stamp is 0)
med = lambda v: float(np.median(v))
print("worker slot (wave 1, step 2): starts %+.2f us after the barrier, tiles %.2f, stores %s; the factor beside it ends at %+.2f, the next barrier at %+.2f" % (
    med([w[0] for w in ws]), med([w[1] for w in ws]), ("%.2f" % med(st)) if st else "not stamped (this leaf form's workers do not store in the slot)",
    med([w[2] for w in ws]), med([w[3] for w in ws])))

# the rows kernel of each step (its first critical workgroup): when it starts running relative to the leaf of the same step, and how long its
# work BEFORE the leaf's flags takes (tile loads + the window's product): if start + pre-work exceeds the leaf's body, the step waits for the rows
rs = [((tl[k][11] - tl[k][1]) / 100, (tl[k][4] - tl[k][11]) / 100, (tl[k][2] - tl[k][1]) / 100) for k in range(1, nblk - 1) if tl[k][11] > 0]
if rs:
    print("rows kernel vs leaf, per step (us): [starts after the leaf's start | its pre-work | the leaf's body]")
    print("  " + "  ".join("%d:[%+.1f|%.1f|%.1f]" % (k + 1, a, b, c) for k, (a, b, c) in enumerate(rs)))

# every step of the factorisation at a glance: start of the leaf (us since the first leaf), time to the next leaf's start, and of that the
# leaf's own wait for its tile -- shows where the update-bound phase (long steps, the chain waiting for the trailing update) ends
st = [us(tl[k][0]) for k in range(nblk)]
print("steps: start | to next | leaf waited")
print("  " + "  ".join("%d:%.0f|%.0f|%.0f" % (k, st[k], st[k + 1] - st[k], (tl[k][1] - tl[k][0]) / 100) for k in range(nblk - 1)))
