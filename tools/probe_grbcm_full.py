"""Where GRBCM.predict(var='full') at config 4's size spends its time (bench leg `grbcm_predict_full`): per-phase wall time with a
synchronisation after each phase -- the local experts' predictions (K* build, V = L^-1 K*, K** - V^T V), the global expert's, and the
aggregation's m x m inversions.  Usage: python tools/probe_grbcm_full.py [m]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import pygpr_amd as pg  # noqa: E402
from pygpr_amd._ops import get_ops, pad_to  # noqa: E402


def synth(n, d, seed):
    rng = np.random.default_rng(seed)
    x = rng.random((n, d))
    return x, np.sin(-x.sum(1)) + 0.1 * rng.standard_normal(n)


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    nc, nl, ng, d = 8, 8192, 1024, 16
    cov = pg.Compose([pg.Squared_exponential(), pg.White_noise()])
    xg, yg = synth(ng, d, 7)
    sh = [synth(nl, d, 100 + c) for c in range(nc)]
    g = pg.GRBCM(torch.from_numpy(np.stack([s[0] for s in sh])), torch.from_numpy(np.stack([s[1] for s in sh])),
                 torch.from_numpy(xg), torch.from_numpy(yg), cov)
    hp = torch.from_numpy(np.concatenate([[1.0], np.full(d, 0.5), [0.1]]))
    g.gpg.set_params(hp)
    g.set_local_params(hp)
    xs = torch.from_numpy(np.random.default_rng(4321).random((m, d))).cuda()
    g.predict(xs, var="full")
    torch.cuda.synchronize()
    ops = get_ops()
    xsd = ops.to_device(xs, g.gpg.dtype)

    def timed(f, reps=3):
        best, out = 1e30, None
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = f()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return 1e3 * best, out

    t_all, _ = timed(lambda: g.predict(xs, var="full"))
    t_g, (mg, vg) = timed(lambda: g.gpg._predict_device(xsd, "full"))
    t_l, (ml, vl) = timed(lambda: g.gpl._predict_device(xsd, "full"))
    t_a, _ = timed(lambda: g._aggregate_full_device(mg[0], vg[0], ml, vl))
    print("m = %d: predict(var='full') %.2f ms = global expert %.2f + %d local experts %.2f + aggregation %.2f" % (m, t_all, t_g, nc, t_l, t_a))
    # one local expert, phase by phase
    e = g.gpl._experts[0]
    from pygpr_amd.covar import spec_of
    spec, _ = spec_of(g.gpl.cov, d)
    m_pad = pad_to(m)
    ks = ops.empty(e.n_pad, m_pad, dtype=g.gpl.dtype)
    v = ops.empty(e.n_pad, m_pad, dtype=g.gpl.dtype)
    c = ops.empty(m_pad, m_pad, dtype=g.gpl.dtype)
    minv = g.gpl._minv(e)
    t1, _ = timed(lambda: ops.kernel_build(spec, e.hp, e.x, xsd, ks))
    t2, _ = timed(lambda: ops.trmm_lower(minv, ks, v))
    t3, _ = timed(lambda: ops.kernel_build(spec, e.hp, xsd, None, c))
    t4, _ = timed(lambda: ops.syrk_tn_sub(v, c, lower_only=True))
    t5, _ = timed(lambda: ops.symmetrize(c, m_pad))
    kt = ops.empty(m_pad, e.n_pad, dtype=g.gpl.dtype)
    vt = ops.empty(m_pad, e.n_pad, dtype=g.gpl.dtype)
    t8, _ = timed(lambda: ops.kernel_build(spec, e.hp, xsd, e.x, kt))
    t9, _ = timed(lambda: ops.trmm_lower_kt(minv, kt, vt))
    c1 = c.reshape(1, m_pad, m_pad)
    t10, _ = timed(lambda: ops.syrk_nt_sub_batched(vt.reshape(1, m_pad, -1), c1))
    vt8 = ops.empty(nc, m_pad, e.n_pad, dtype=g.gpl.dtype)
    vt8[:] = vt
    c8 = ops.zeros(nc, m_pad, m_pad, dtype=g.gpl.dtype)
    t11, _ = timed(lambda: ops.syrk_nt_sub_batched(vt8, c8))
    minvs = [g.gpl._minv(ee) for ee in g.gpl._experts]
    kt8 = ops.empty(nc, m_pad, e.n_pad, dtype=g.gpl.dtype)
    kt8[:] = kt
    t12, _ = timed(lambda: ops.trmm_lower_kt(minvs, kt8, vt8))
    print("Vt of %d experts in one launch %.3f ms (%.1f TFLOP/s)" % (nc, t12, nc * float(e.n_pad) ** 2 * m_pad / t12 / 1e9))
    n = e.n_pad
    print("test-point-major forms: K*^T %.3f, Vt = K* L^-T %.3f (%.1f TFLOP/s), rank-n update of one expert %.3f, of %d experts in one launch %.3f "
          "(%.1f TFLOP/s)" % (t8, t9, float(n) ** 2 * m_pad / t9 / 1e9, t10, nc, t11, nc * float(n) * m_pad ** 2 / t11 / 1e9))
    print("one local expert (n = %d): K* %.3f, V = L^-1 K* %.3f (%.1f TFLOP/s), K** %.3f, K** - V^T V %.3f (%.1f TFLOP/s), mirror %.3f ms"
          % (n, t1, t2, float(n) ** 2 * m_pad / t2 / 1e9, t3, t4, float(n) * m_pad ** 2 / t4 / 1e9, t5))
    a = g._padded_spd(vl[0])
    t6, _ = timed(lambda: ops.spd_inverse_lower(a.clone()))
    t7, _ = timed(lambda: a.clone())
    print("one m x m inversion (spd_inverse_lower, m_pad = %d): %.3f ms (the copy alone %.3f)" % (a.shape[0], t6, t7))


if __name__ == "__main__":
    main()
