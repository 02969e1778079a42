"""Pins the CPU oracle (oracle/pygpr_oracle.py) to the golden vectors captured from the
reference import (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import pygpr_oracle as orc

COV = {"se": orc.SE, "wn": orc.WN}


def covs_of(spec):
    return [COV[s] for s in str(spec).split(",")]


def test_covar_kernels_match_reference(golden):
    g = golden("covar")
    for i in range(int(g["ncase"])):
        p = "c%02d_" % i
        covs = covs_of(g[p + "spec"])
        x, xp, hp = g[p + "x"], g[p + "xp"], g[p + "hp"]
        batched = x.ndim == 3
        for c in range(x.shape[0] if batched else 1):
            xc, hc = (x[c], hp[c]) if batched else (x, hp)
            k_ref = g[p + "k"][c] if batched else g[p + "k"]
            dk_ref = g[p + "dk"][c] if batched else g[p + "dk"]
            ks_ref = g[p + "ks"]
            k, dk = orc.kernel_and_grad(covs, hc, xc)
            np.testing.assert_allclose(k, k_ref, rtol=0, atol=1e-14)
            np.testing.assert_allclose(dk, dk_ref, rtol=0, atol=1e-13)
            np.testing.assert_allclose(orc.kernel(covs, hc, xc), k_ref, rtol=0, atol=1e-14)
            ks = orc.kernel(covs, hc, xc, xp)
            if ks_ref.ndim == 0:  # White_noise alone with xp -> int tensor(0) (covar.py:243)
                assert int(ks) == 0
            else:
                np.testing.assert_allclose(ks, ks_ref[c] if batched else ks_ref, rtol=0, atol=1e-14)
            # direct-difference distances (the HIP formulation) agree to rounding
            kd = orc.kernel(covs, hc, xc, form="direct")
            np.testing.assert_allclose(kd, k_ref, rtol=0, atol=1e-13)


def test_exact_gp_small(golden):
    g = golden("gp")
    covs = [orc.SE, orc.WN]
    x, y, xp, hp = g["a_x"], g["a_y"], g["a_xp"], g["a_hp"]
    k, chol, alpha = orc.gp_update(covs, hp, x, y)
    np.testing.assert_allclose(k, g["a_krn"], atol=1e-14)
    np.testing.assert_allclose(chol, g["a_chol"], atol=1e-12)
    np.testing.assert_allclose(alpha, g["a_wt"], rtol=1e-9)
    mu, var = orc.gp_predict(covs, hp, x, y, xp, "diag")
    np.testing.assert_allclose(mu, g["a_mu"], atol=1e-11)
    np.testing.assert_allclose(var, g["a_var"], atol=1e-12)
    mu, cov = orc.gp_predict(covs, hp, x, y, xp, "full")
    np.testing.assert_allclose(cov, g["a_cov"], atol=1e-12)
    np.testing.assert_allclose(mu, g["a_sk_mu"], atol=1e-11)  # SK_WRAP.predict == mean


@pytest.mark.parametrize("route", ["solve", "kinv"])
def test_mle_small(golden, route):
    g = golden("gp")
    covs = [orc.SE, orc.WN]
    x, y, hp = g["a_x"], g["a_y"], g["a_hp"]
    np.testing.assert_allclose(orc.mle_loss(covs, hp, x, y), g["a_loss"], rtol=1e-12)
    loss, grad = orc.mle_loss_and_grad(covs, hp, x, y, route)
    np.testing.assert_allclose(loss, g["a_loss2"], rtol=1e-12)
    np.testing.assert_allclose(grad, g["a_grad2"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(grad, g["a_grad"], rtol=1e-9, atol=1e-9)
    # default hp: sigma_n = 1e-4 -> cond(K) ~ 1e9, tolerance scales with it
    loss, grad = orc.mle_loss_and_grad(covs, g["a_hp0"], x, y, route)
    np.testing.assert_allclose(loss, g["a_loss0"], rtol=1e-8)
    np.testing.assert_allclose(grad, g["a_grad0"], rtol=1e-5, atol=1e-5 * np.abs(g["a_grad0"]).max())


def test_known_answer_cfg1(golden):
    """n=512, d=2, default hp -> NLML = -3461.42170686."""
    g = golden("gp")
    covs = [orc.SE, orc.WN]
    assert abs(float(g["b_loss0"]) - (-3461.42170686)) < 1e-6
    for route in ("solve", "kinv"):
        loss, grad = orc.mle_loss_and_grad(covs, g["b_hp0"], g["b_x"], g["b_y"], route)
        np.testing.assert_allclose(loss, g["b_loss0"], rtol=1e-8)
        np.testing.assert_allclose(grad, g["b_grad0"], rtol=1e-5)
        loss, grad = orc.mle_loss_and_grad(covs, g["b_hp"], g["b_x"], g["b_y"], route)
        np.testing.assert_allclose(loss, g["b_loss"], rtol=1e-10)
        np.testing.assert_allclose(grad, g["b_grad"], rtol=1e-8, atol=1e-8 * np.abs(g["b_grad"]).max())
    mu, var = orc.gp_predict(covs, g["b_hp"], g["b_x"], g["b_y"], g["b_xp"], "diag")
    np.testing.assert_allclose(mu, g["b_mu"], atol=1e-10)
    np.testing.assert_allclose(var, g["b_var"], atol=1e-11)
    # the lean CPU baseline (what bench.py times) is the same function
    loss, grad = orc.mle_loss_and_grad_lean(g["b_hp"], g["b_x"], g["b_y"])
    np.testing.assert_allclose(loss, g["b_loss"], rtol=1e-10)
    np.testing.assert_allclose(grad, g["b_grad"], rtol=1e-8, atol=1e-8 * np.abs(g["b_grad"]).max())


def test_batched_experts(golden):
    g = golden("gp")
    covs = [orc.SE, orc.WN]
    for c in range(g["c_x"].shape[0]):
        x, y, hp = g["c_x"][c], g["c_y"][c], g["c_hp"][c]
        _, chol, alpha = orc.gp_update(covs, hp, x, y)
        np.testing.assert_allclose(chol, g["c_chol"][c], atol=1e-12)
        np.testing.assert_allclose(alpha, g["c_wt"][c], rtol=1e-9)
        mu, var = orc.gp_predict(covs, hp, x, y, g["c_xp"], "diag")
        np.testing.assert_allclose(mu, g["c_mu"][c], atol=1e-11)
        np.testing.assert_allclose(var, g["c_var"][c], atol=1e-12)
        _, cov = orc.gp_predict(covs, hp, x, y, g["c_xp"], "full")
        np.testing.assert_allclose(cov, g["c_cov"][c], atol=1e-12)
        loss, grad = orc.mle_loss_and_grad(covs, hp, x, y, "kinv")
        np.testing.assert_allclose(loss, g["c_loss"][c], rtol=1e-12)
        np.testing.assert_allclose(grad, g["c_grad"][c], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(loss, g["c_lossonly"][c], rtol=1e-12)
        np.testing.assert_allclose(grad, g["c_gradonly"][c], rtol=1e-9, atol=1e-9)


def test_learn_rate(golden):
    g = golden("gp")
    gam = orc.get_learn_rate([orc.SE, orc.WN], g["a_hp"], g["a_x"], g["a_y"], 1e-6)
    # second difference with eps = 1e-6: rounding in the three losses is amplified by 1 / eps^2 (measured 3e-10 against the reference)
    np.testing.assert_allclose(gam, g["a_gamma"], rtol=1e-7)
    r3 = golden("round3")          # round 3: the same probe at step sizes above the rounding floor (make_golden_r3.py)
    for tag in "34":
        gam = orc.get_learn_rate([orc.SE, orc.WN], g["a_hp"], g["a_x"], g["a_y"], float(r3["lr_eps" + tag]))
        np.testing.assert_allclose(gam, r3["lr_gamma" + tag], rtol=1e-11)


def test_reference_test_sizes(golden):
    """The reference's own test sizes (tests/test_gpr.py:59-100: nc = 10 experts of n = 100 points; tests/test_grbcm.py:18-37:
    nc = 5, ng = 100, nls = 50): the oracle against what the imported reference returned (round3.npz)."""
    r = golden("round3")
    covs = [orc.SE, orc.WN]
    for c in range(r["sm_x"].shape[0]):
        x, y, hp = r["sm_x"][c], r["sm_y"][c], r["sm_hp"][c]
        mu, var = orc.gp_predict(covs, hp, x, y, r["sm_xp"], "diag")
        np.testing.assert_allclose(mu, r["sm_mu"][c], atol=1e-10)
        np.testing.assert_allclose(var, r["sm_var"][c], atol=1e-11)
        loss, grad = orc.mle_loss_and_grad(covs, hp, x, y, "kinv")
        np.testing.assert_allclose(loss, r["sm_loss"][c], rtol=1e-11)
        np.testing.assert_allclose(grad, r["sm_grad"][c], rtol=1e-8, atol=1e-8 * np.abs(r["sm_grad"][c]).max())
    nc = r["sg_xl"].shape[0]
    mu, var, beta, prec = orc.grbcm_predict(covs, r["sg_hp"], np.tile(r["sg_hp"], (nc, 1)), r["sg_xl"], r["sg_yl"], r["sg_xg"],
                                            r["sg_yg"], r["sg_xl"][2], "diag")
    np.testing.assert_allclose(mu, r["sg_mu"], atol=1e-9)
    np.testing.assert_allclose(var, r["sg_var"], rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(beta, r["sg_beta"], rtol=1e-7, atol=1e-9)


def test_grbcm(golden):
    g = golden("grbcm")
    covs = [orc.SE, orc.WN]
    for i in range(int(g["ncase"])):
        p = "g%d_" % i
        mu, var, beta, prec = orc.grbcm_predict(covs, g[p + "hpg"], g[p + "hpl"], g[p + "xl"], g[p + "yl"],
                                                g[p + "xg"], g[p + "yg"], g[p + "xs"], "diag")
        np.testing.assert_allclose(mu, g[p + "mu"], atol=1e-10)
        np.testing.assert_allclose(var, g[p + "var"], atol=1e-11)
        np.testing.assert_allclose(beta, g[p + "beta"], atol=1e-9)
        np.testing.assert_allclose(prec, g[p + "prec"], rtol=1e-9)
        # the rank-local decomposition used by the multi-GPU path gives the same answer
        x, y = orc.grbcm_data(g[p + "xl"], g[p + "yl"], g[p + "xg"], g[p + "yg"])
        mg, vg = orc.gp_predict(covs, g[p + "hpg"], g[p + "xg"], g[p + "yg"], g[p + "xs"], "diag")
        sums = 0.0
        for c in range(x.shape[0]):
            mc, vc = orc.gp_predict(covs, g[p + "hpl"][c], x[c], y[c], g[p + "xs"], "diag")
            sums = sums + orc.grbcm_terms(mc, vc, vg, c == 0)
        mu2, var2 = orc.grbcm_finish(sums, mg, vg)
        np.testing.assert_allclose(mu2, g[p + "mu"], atol=1e-10)
        np.testing.assert_allclose(var2, g[p + "var"], atol=1e-11)
        if int(g[p + "full_ok"]):
            hpl = np.broadcast_to(g[p + "hpg"], g[p + "hpl"].shape)
            mu_f, cov_f = orc.grbcm_predict(covs, g[p + "hpg"], hpl, g[p + "xl"], g[p + "yl"],
                                            g[p + "xg"], g[p + "yg"], g[p + "xs"], "full")
            np.testing.assert_allclose(cov_f, g[p + "cov_full"], rtol=1e-6, atol=1e-10)
            np.testing.assert_allclose(mu_f, g[p + "mu_full"], rtol=1e-6, atol=1e-9)


def test_matern52_unpinned_by_reference_matches_sklearn():
    """Matern-5/2 has no reference oracle: pin K to sklearn and dK to FD."""
    from sklearn.gaussian_process.kernels import Matern

    rng = np.random.default_rng(3)
    x, xp = rng.random((30, 4)), rng.random((5, 4))
    hp = np.array([1.3, 0.7, 1.1, 0.9, 1.4])
    k = orc.matern52_kernel(hp, x)
    ksk = hp[0] ** 2 * Matern(length_scale=1.0 / hp[1:], nu=2.5)(x)
    np.testing.assert_allclose(k, ksk, atol=1e-13)
    ks = orc.matern52_kernel(hp, x, xp)
    np.testing.assert_allclose(ks, hp[0] ** 2 * Matern(length_scale=1.0 / hp[1:], nu=2.5)(xp, x), atol=1e-13)
    _, dk = orc.matern52_kernel_and_grad(hp, x)
    for a in range(hp.size):
        e = np.zeros_like(hp)
        e[a] = 1e-6
        fd = (orc.matern52_kernel(hp + e, x) - orc.matern52_kernel(hp - e, x)) / 2e-6
        np.testing.assert_allclose(dk[a], fd, atol=1e-8)


def test_matern52_lean_nlml_is_the_oracles_nlml():
    """`matern52_nlml_lean` (the full-size config-5 check of tests/test_parity_gpu.py evaluates it at n = 33792, where the [n, n, d]
    difference array of `matern52_kernel` does not fit) is `mle_loss([M52, WN])` to rounding: row slabs that do and do not divide n."""
    x, y = orc.synth(700, 16, seed=9)
    hp = np.concatenate([[1.1], 0.3 + 0.4 * np.random.default_rng(2).random(16), [0.1]])
    ref = orc.mle_loss([orc.M52, orc.WN], hp, x, y)
    for rows in (128, 333, 1024):
        np.testing.assert_allclose(orc.matern52_nlml_lean(hp, x, y, rows=rows), ref, rtol=1e-12)


def test_as_written_torch_baseline_matches_solve_route():
    """bench.py's second CPU baseline (torch substrate) is the same function as the as-written oracle route."""
    rng = np.random.default_rng(5)
    n, d = 96, 3
    x = rng.random((n, d))
    y = np.sin(-x.sum(1)) + 0.1 * rng.standard_normal(n)
    hp = np.concatenate([[1.3], 0.5 + rng.random(d), [0.1]])
    f0, g0 = orc.mle_loss_and_grad([orc.SE(), orc.WN()], hp, x, y, route="solve")
    f1, g1 = orc.mle_loss_and_grad_as_written(hp, x, y)
    f2, g2 = orc.mle_loss_and_grad_lean(hp, x, y)
    assert abs(f0 - f1) <= 1e-11 * abs(f0) and abs(f0 - f2) <= 1e-11 * abs(f0)
    np.testing.assert_allclose(g1, g0, rtol=1e-9, atol=1e-9 * np.abs(g0).max())
    np.testing.assert_allclose(g2, g0, rtol=1e-9, atol=1e-9 * np.abs(g0).max())


def test_oracle_round2_surface_fixtures(golden):
    """Pins the oracle to the round-2 fixtures (make_golden_r2.py): distance, an 11-child Compose, batched test points,
    batched params on shared points."""
    g = golden("surface2")
    np.testing.assert_allclose(orc.se_distance(g["d_x"]), g["d_sq"], atol=1e-15)
    np.testing.assert_allclose(orc.se_distance(g["d_x"], g["d_xp"]), g["d_sqp"], atol=1e-15)
    np.testing.assert_allclose(orc.se_distance(g["d_x"], g["d_xp"], form="direct"), g["d_sqp"], atol=1e-14)
    for c in range(3):
        np.testing.assert_allclose(orc.se_distance(g["d_xb"][c], g["d_xpb"][c]), g["d_sqpb"][c], atol=1e-15)
    covs = covs_of(g["L_kinds"])
    k, dk = orc.kernel_and_grad(covs, g["L_hp"], g["L_x"])
    np.testing.assert_allclose(k, g["L_k"], atol=1e-14)
    np.testing.assert_allclose(dk, g["L_dk"], atol=1e-13)
    np.testing.assert_allclose(orc.kernel(covs, g["L_hp"], g["L_x"], g["L_xp"]), g["L_ks"], atol=1e-14)
    mu, var = orc.gp_predict(covs, g["L_hp"], g["L_x"], g["L_y"], g["L_xp"], "diag")
    np.testing.assert_allclose(mu, g["L_mu"], atol=1e-11)
    np.testing.assert_allclose(var, g["L_var"], atol=1e-12)
    for route in ("solve", "kinv"):
        l, gr = orc.mle_loss_and_grad(covs, g["L_hp"], g["L_x"], g["L_y"], route)
        np.testing.assert_allclose(l, g["L_loss"], rtol=1e-11)
        np.testing.assert_allclose(gr, g["L_grad"], rtol=1e-8, atol=1e-8 * np.abs(g["L_grad"]).max())
    sw = [orc.SE, orc.WN]
    for c in range(3):
        mu, var = orc.gp_predict(sw, g["bx_hp"][c], g["bx_x"][c], g["bx_y"][c], g["bx_xp"][c], "diag")
        np.testing.assert_allclose(mu, g["bx_mu"][c], atol=1e-11)
        np.testing.assert_allclose(var, g["bx_var"][c], atol=1e-12)
        _, cov = orc.gp_predict(sw, g["bx_hp"][c], g["bx_x"][c], g["bx_y"][c], g["bx_xp"][c], "full")
        np.testing.assert_allclose(cov, g["bx_cov"][c], atol=1e-12)
    for c in range(4):
        mu, var = orc.gp_predict(sw, g["bp_hp"][c], g["bp_x"], g["bp_y"], g["bp_xp"], "diag")
        np.testing.assert_allclose(mu, g["bp_mu"][c], atol=1e-11)
        np.testing.assert_allclose(orc.mle_loss(sw, g["bp_hp"][c], g["bp_x"], g["bp_y"]), g["bp_loss"][c], rtol=1e-11)
    _, chol, _ = orc.gp_update([orc.SE], g["r1_hp"], g["r1_x"], g["r1_y"])     # the rank-one covariance factorises
    np.testing.assert_allclose(chol, g["r1_krnchd"], rtol=1e-7, atol=1e-12)
