"""GPU parity of the PyGPR-compatible class surface: golden vectors captured from the reference
(tests/golden/*.npz), the CPU oracle on seeded inputs, and the reference's own property tests restated
(PyGPR/tests/test_covar.py, test_gpr.py, test_loss.py, test_grbcm.py) with fixed seeds.

Stated fp64 tolerances: with sigma_n = 0.1 NLML rtol 1e-10, gradient rtol 1e-8 on |g|_inf,
mean atol 1e-10, variance atol 1e-11; default hp (sigma_n = 1e-4, cond(K) ~ 1e9): NLML 1e-8, gradient 1e-7 (SURVEY: 1e-6;
measured 2.8e-9 at n = 512, cond(K) = 3.5e9 -- test_default_hp_gradient_error_is_stated prints it)."""
import os

import numpy as np
import pytest
import torch

import pygpr_amd as pg
from oracle import pygpr_oracle as orc

pytestmark = pytest.mark.gpu

COV = {"se": pg.Squared_exponential, "wn": pg.White_noise}
OCOV = {"se": orc.SE, "wn": orc.WN}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def N(t):
    return t.detach().cpu().numpy()


def make_cov(spec):
    names = str(spec).split(",")
    return COV[names[0]]() if len(names) == 1 else pg.Compose([COV[s]() for s in names])


def se_wn():
    return pg.Compose([pg.Squared_exponential(), pg.White_noise()])


@pytest.fixture(autouse=True)
def _scratch_cwd(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)  # optimisers write opt.dat into cwd (opt.py:48)


# ------------------------------------------------------------------ covariance kernels
def test_covar_golden(golden):
    g = golden("covar")
    for i in range(int(g["ncase"])):
        p = "c%02d_" % i
        cov = make_cov(g[p + "spec"])
        x, xp, hp = T(g[p + "x"]), T(g[p + "xp"]), T(g[p + "hp"])
        k = cov.kernel(hp, x)
        assert k.shape == g[p + "k"].shape and k.dtype == torch.float64 and k.device.type == "cpu"
        np.testing.assert_allclose(N(k), g[p + "k"], atol=1e-13)
        k2, dk = cov.kernel_and_grad(hp, x)
        np.testing.assert_allclose(N(k2), g[p + "k"], atol=1e-13)
        assert dk.shape == g[p + "dk"].shape
        np.testing.assert_allclose(N(dk), g[p + "dk"], atol=1e-12)
        ks = cov.kernel(hp, x, xp)
        if g[p + "ks"].ndim == 0:
            assert ks.dim() == 0 and int(ks) == 0 and ks.dtype == torch.int64   # covar.py:243
        else:
            assert ks.shape == g[p + "ks"].shape
            np.testing.assert_allclose(N(ks), g[p + "ks"], atol=1e-13)
        assert torch.equal(hp, T(g[p + "hp"])) and torch.equal(x, T(g[p + "x"]))   # callers' tensors untouched


def test_covar_protocol_shapes_and_errors():
    x = torch.rand(4, 10, 3, dtype=torch.float64)
    cov = pg.Compose([pg.Squared_exponential(), pg.Squared_exponential(), pg.White_noise()])
    assert cov.get_params_shape(x) == [4, 9]
    hp = cov.init_params(x)
    assert hp.shape == (4, 9) and float(hp[0, -1]) == 1e-4 and float(hp[0, 0]) == 1.0
    with pytest.raises(AssertionError):
        cov.kernel(torch.ones(4, 8, dtype=torch.float64), x)
    # hp [nc, .] against unbatched x broadcasts to [nc, n, n]
    k = pg.Squared_exponential().kernel(torch.rand(3, 4, dtype=torch.float64) + 0.5, x[0])
    assert k.shape == (3, 10, 10)


@pytest.mark.parametrize("n,d", [(10, 2), (100, 5), (1000, 2)])
def test_covar_properties(n, d):
    """test_covar.py:90-163 restated: symmetric, PD with jitter, analytic dK vs forward FD."""
    torch.manual_seed(n + d)
    cov = pg.Compose([pg.Squared_exponential(), pg.Squared_exponential(), pg.White_noise()])
    x = torch.rand(n, d, dtype=torch.float64)
    hp = torch.rand(2 * d + 3, dtype=torch.float64)
    k, dk = cov.kernel_and_grad(hp, x)
    assert torch.equal(k, k.T)
    assert torch.linalg.eigvalsh(k + 1e-7 * torch.eye(n, dtype=torch.float64)).min() > 0
    eps = 1e-5
    for p in range(hp.numel()):
        hp2 = hp.clone()
        hp2[p] += eps
        fd = (cov.kernel(hp2, x) - k) / eps
        assert torch.allclose(dk[p], fd, atol=1e-4)


def test_matern52_against_oracle():
    rng = np.random.default_rng(3)
    x, xp = rng.random((70, 4)), rng.random((9, 4))
    hp = np.array([1.3, 0.7, 1.1, 0.9, 1.4])
    cov = pg.Matern52()
    np.testing.assert_allclose(N(cov.kernel(T(hp), T(x))), orc.matern52_kernel(hp, x), atol=1e-13)
    np.testing.assert_allclose(N(cov.kernel(T(hp), T(x), T(xp))), orc.matern52_kernel(hp, x, xp), atol=1e-13)
    k, dk = cov.kernel_and_grad(T(hp), T(x))
    np.testing.assert_allclose(N(dk), orc.matern52_kernel_and_grad(hp, x)[1], atol=1e-12)


# ------------------------------------------------------------------ Exact_GP
def test_exact_gp_golden(golden):
    g = golden("gp")
    gp = pg.Exact_GP(T(g["a_x"]), T(g["a_y"]), se_wn())
    assert gp.need_upd and torch.equal(gp.params, T(g["a_hp0"]))
    gp.set_params(T(g["a_hp"]))
    mu, var = gp.predict(T(g["a_xp"]), var="diag")
    assert not gp.need_upd
    np.testing.assert_allclose(N(mu), g["a_mu"], atol=1e-10)
    np.testing.assert_allclose(N(var), g["a_var"], atol=1e-11)
    mu, cov = gp.predict(T(g["a_xp"]), var="full")
    np.testing.assert_allclose(N(cov), g["a_cov"], atol=1e-11)
    assert torch.equal(cov, cov.T)
    np.testing.assert_allclose(N(gp.krn), g["a_krn"], atol=1e-13)
    np.testing.assert_allclose(N(gp.krnchd), g["a_chol"], atol=1e-11)
    np.testing.assert_allclose(N(gp.wt), g["a_wt"], rtol=1e-8)
    mu, none = gp.predict(T(g["a_xp"]), var="none")
    assert none is NotImplemented
    # SK_WRAP: fit + predict + score (scikit_model.py:15-35)
    sk = pg.SK_WRAP(gp).fit(T(g["a_x"]), T(g["a_y"]))
    np.testing.assert_allclose(N(sk.predict(T(g["a_xp"]))), g["a_sk_mu"], atol=1e-10)
    np.testing.assert_allclose(sk.score(T(g["a_x"]), T(g["a_y"])), g["a_sk_score"], rtol=1e-8)


def test_exact_gp_cfg1_and_batched(golden):
    g = golden("gp")
    for eager in (False, True):                    # alpha by blocked substitution / by L^-T (L^-1 y): same answers
        gpe = pg.Exact_GP(T(g["a_x"]), T(g["a_y"]), se_wn(), eager_inverse=eager)
        gpe.set_params(T(g["a_hp"]))
        mu, var = gpe.predict(T(g["a_xp"]), var="diag")
        np.testing.assert_allclose(N(mu), g["a_mu"], atol=1e-10)
        np.testing.assert_allclose(N(var), g["a_var"], atol=1e-11)
        np.testing.assert_allclose(N(gpe.wt), g["a_wt"], rtol=1e-8)
    gp = pg.Exact_GP(T(g["b_x"]), T(g["b_y"]), se_wn())
    gp.set_params(T(g["b_hp"]))
    mu, var = gp.predict(T(g["b_xp"]), var="diag")
    np.testing.assert_allclose(N(mu), g["b_mu"], atol=1e-10)
    np.testing.assert_allclose(N(var), g["b_var"], atol=1e-11)
    # batched experts
    gp = pg.Exact_GP(T(g["c_x"]), T(g["c_y"]), se_wn())
    gp.set_params(T(g["c_hp"]))
    mu, var = gp.predict(T(g["c_xp"]), var="diag")
    assert mu.shape == g["c_mu"].shape and var.shape == g["c_var"].shape
    np.testing.assert_allclose(N(mu), g["c_mu"], atol=1e-10)
    np.testing.assert_allclose(N(var), g["c_var"], atol=1e-11)
    _, cov = gp.predict(T(g["c_xp"]), var="full")
    np.testing.assert_allclose(N(cov), g["c_cov"], atol=1e-11)
    np.testing.assert_allclose(N(gp.krnchd), g["c_chol"], atol=1e-11)
    np.testing.assert_allclose(N(gp.wt), g["c_wt"], rtol=1e-8)
    # squeeze_() semantics for a single test point (SURVEY a-9 / a-10)
    mu1, var1 = gp.predict(T(g["c_xp"][:1]), var="diag")
    assert mu1.shape == (4,) and var1.shape == (4, 1)


@pytest.mark.parametrize("n,nc", [(100, None), (50, 5)])
def test_exact_gp_properties(n, nc):
    """test_gpr.py:17-100 restated: interpolation of y = sin(-sum x), covariance symmetric PSD."""
    torch.manual_seed(n)
    d = 3
    x = torch.rand((n, d) if nc is None else (nc, n, d), dtype=torch.float64)
    y = torch.sin(-x.sum(-1))
    gp = pg.Exact_GP(x, y, se_wn())
    xs = x if nc is None else x[0]
    mu, cov = gp.predict(xs, var="full")
    target = y if nc is None else y[0]
    assert torch.allclose(mu if nc is None else mu[0], target, atol=1e-4)
    c0 = cov if nc is None else cov[0]
    assert torch.allclose(c0, c0.T, atol=1e-7)
    assert torch.linalg.eigvalsh(c0).min() > -1e-7
    if nc is not None:
        assert mu.shape == (nc, n) and cov.shape == (nc, n, n)


def test_not_positive_definite_raises():
    """torch.cholesky raises torch.linalg.LinAlgError (a RuntimeError) naming the failing leading minor
    (SURVEY 5 / 8b); a NaN hyper-parameter makes the very first pivot fail.  (The rank-one covariance round 1 first
    tried here is factorised by the reference too -- fixture r1_* of surface2.npz, check_rank_one_covariance.)"""
    x = torch.rand(40, 2, dtype=torch.float64)
    gp = pg.Exact_GP(x, torch.rand(40, dtype=torch.float64), se_wn())
    gp.set_params(torch.tensor([float("nan"), 1.0, 1.0, 0.1], dtype=torch.float64))
    with pytest.raises(torch.linalg.LinAlgError, match="leading minor of order 1 "):
        gp.update()
    with pytest.raises(RuntimeError):
        pg.MLE(gp).loss_and_grad(np.array([float("nan"), 1.0, 1.0, 0.1]))
    gp.set_params(torch.tensor([1.0, 1.0, 1.0, 0.1], dtype=torch.float64))
    gp.update()                                   # the model recovers with valid parameters


# ------------------------------------------------------------------ MLE
def test_mle_golden(golden):
    g = golden("gp")
    gp = pg.Exact_GP(T(g["a_x"]), T(g["a_y"]), se_wn())
    mle = pg.MLE(gp)
    loss = mle.loss(g["a_hp"].copy())
    assert isinstance(loss, np.ndarray) and loss.ndim == 0
    np.testing.assert_allclose(loss, g["a_loss"], rtol=1e-10)
    grad = mle.grad(g["a_hp"].copy())
    np.testing.assert_allclose(grad, g["a_grad"], rtol=1e-8, atol=1e-8 * np.abs(g["a_grad"]).max())
    l2, g2 = mle.loss_and_grad(g["a_hp"].copy())
    np.testing.assert_allclose(l2, g["a_loss2"], rtol=1e-10)
    np.testing.assert_allclose(g2, g["a_grad2"], rtol=1e-8, atol=1e-8 * np.abs(g["a_grad2"]).max())
    assert mle.loss_value is l2 and mle.grad_value is g2
    l3, g3 = mle.loss_and_grad(g["a_hp"].copy())          # memoised repeat: same numbers without a device evaluation
    assert float(l3) == float(l2) and np.array_equal(g3, g2)
    l0, g0 = mle.loss_and_grad(g["a_hp0"].copy())                      # cond(K) ~ 1e9
    np.testing.assert_allclose(l0, g["a_loss0"], rtol=1e-8)
    # SURVEY 8c states 1e-6 for this class; measured 1.4e-10 (n = 64) and 2.8e-9 (n = 512): test_default_hp_gradient_error_is_stated
    np.testing.assert_allclose(g0, g["a_grad0"], rtol=1e-7, atol=1e-7 * np.abs(g["a_grad0"]).max())
    # cfg1 known answer
    gpb = pg.Exact_GP(T(g["b_x"]), T(g["b_y"]), se_wn())
    lb, gb = pg.MLE(gpb).loss_and_grad(g["b_hp"].copy())
    np.testing.assert_allclose(lb, g["b_loss"], rtol=1e-10)
    np.testing.assert_allclose(gb, g["b_grad"], rtol=1e-8, atol=1e-8 * np.abs(g["b_grad"]).max())
    lb0, gb0 = pg.MLE(gpb).loss_and_grad(g["b_hp0"].copy())
    np.testing.assert_allclose(lb0, -3461.42170686, rtol=1e-8)
    np.testing.assert_allclose(gb0, g["b_grad0"], rtol=1e-7, atol=1e-7 * np.abs(g["b_grad0"]).max())
    # batched [nc, nhp]
    gpc = pg.Exact_GP(T(g["c_x"]), T(g["c_y"]), se_wn())
    lc, gc = pg.MLE(gpc).loss_and_grad(g["c_hp"].copy())
    assert lc.shape == g["c_loss"].shape and gc.shape == g["c_grad"].shape
    np.testing.assert_allclose(lc, g["c_loss"], rtol=1e-10)
    np.testing.assert_allclose(gc, g["c_grad"], rtol=1e-8, atol=1e-8 * np.abs(g["c_grad"]).max())
    np.testing.assert_allclose(pg.MLE(gpc).loss(g["c_hp"].copy()), g["c_lossonly"], rtol=1e-10)


@pytest.mark.parametrize("n,d", [(10, 2), (100, 3), (1000, 5), (100, 7)])
def test_mle_grad_vs_fd(n, d):
    """test_loss.py:17-44 restated (forward FD of MLE.loss); central differences, looser eps."""
    torch.manual_seed(n * d)
    x = torch.rand(n, d, dtype=torch.float64)
    y = torch.sin(-x.sum(-1))
    mle = pg.MLE(pg.Exact_GP(x, y, se_wn()))
    hp = np.random.default_rng(n).random(d + 2) * 0.5 + 0.5
    hp[-1] = 0.05 + 0.05 * hp[-1]
    grad = mle.grad(hp.copy())
    fd = np.empty_like(hp)
    for p in range(hp.size):
        e = np.zeros_like(hp)
        e[p] = 1e-6
        fd[p] = (mle.loss(hp + e) - mle.loss(hp - e)) / 2e-6
    assert np.max(np.abs(fd - grad)) < 1e-3 * max(1.0, np.abs(grad).max())


def test_learn_rate_and_cg(golden):
    g = golden("gp")
    gp = pg.Exact_GP(T(g["a_x"]), T(g["a_y"]), se_wn())
    gam = pg.get_learn_rate(T(g["a_hp"]), pg.MLE(gp), 1e-6)
    # eps = 1e-6: a second difference of three NLML values whose rounding (1e-14 relative) is amplified by 1 / eps^2
    np.testing.assert_allclose(gam, g["a_gamma"], rtol=1e-6)     # measured 8.6e-10
    r3 = golden("round3")        # the same probe at step sizes above the rounding floor (make_golden_r3.py); measured 1e-14 / 6e-14
    for tag, rtol in (("3", 1e-10), ("4", 1e-9)):
        gam = pg.get_learn_rate(T(g["a_hp"]), pg.MLE(gp), float(r3["lr_eps" + tag]))
        np.testing.assert_allclose(gam, r3["lr_gamma" + tag], rtol=rtol)
    gpd = pg.Exact_GP(T(g["d_x"]), T(g["d_y"]), se_wn())
    gpd.set_params(T(g["d_hp"]))
    cg = pg.CG(pg.MLE(gpd))
    cg.args["maxiter"] = 5
    cg.args["disp"] = False
    cg.minimize()
    assert os.path.exists("opt.dat")
    assert int(cg.res.nit) == int(g["d_nit"])
    np.testing.assert_allclose(cg.res.fun, g["d_res_fun"], rtol=1e-6)
    np.testing.assert_allclose(cg.res.x, g["d_res_x"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(N(gpd.params), g["d_params"], rtol=1e-4, atol=1e-5)
    assert gpd.need_upd


def test_quadratic_optimisers_on_a_mock_loss():
    """test_opt.py:20-56 restated: CG_Quad / BFGS_Quad reach solve(H, -J) on a random SPD quadratic."""
    rng = np.random.default_rng(5)
    dim = 6
    a = rng.standard_normal((dim, dim))
    h = a @ a.T + dim * np.eye(dim)
    j = rng.standard_normal(dim)
    for cls in (pg.CG_Quad, pg.BFGS_Quad):
        loss = pg.Loss(None)
        loss.loss = lambda x: 0.5 * x @ h @ x + j @ x
        loss.grad = lambda x: h @ x + j
        opt = cls(loss, gtol=1e-8, max_iter=200)
        opt.minimize(par=np.zeros(dim))
        assert np.all(np.isclose(opt.x, np.linalg.solve(h, -j), rtol=1e-3))
    np.testing.assert_allclose(pg.hessian(np.zeros(dim), lambda x: h @ x + j, 1e-6), h, rtol=1e-5)


# ------------------------------------------------------------------ grBCM
def test_grbcm_golden(golden):
    g = golden("grbcm")
    for i in range(int(g["ncase"])):
        p = "g%d_" % i
        m = pg.GRBCM(T(g[p + "xl"]), T(g[p + "yl"]), T(g[p + "xg"]), T(g[p + "yg"]), se_wn())
        m.gpg.set_params(T(g[p + "hpg"]))
        m.gpl.set_params(T(g[p + "hpl"]))
        mu, var = m.predict(T(g[p + "xs"]), var="diag")
        np.testing.assert_allclose(N(mu), g[p + "mu"], atol=1e-10)
        np.testing.assert_allclose(N(var), g[p + "var"], atol=1e-11)
        np.testing.assert_allclose(N(m.beta), g[p + "beta"], atol=1e-9)
        np.testing.assert_allclose(N(m.prec), g[p + "prec"], rtol=1e-9)
        assert (m.nc, m.nsc, m.ng, m.dim) == (g[p + "xl"].shape[0], 50, 20, 3)
        if int(g[p + "full_ok"]):   # full-covariance committee (gr_bcm.py:99-114), shared hp as in the fixture
            m.gpl.set_params(T(np.broadcast_to(g[p + "hpg"], g[p + "hpl"].shape).copy()))
            mu_f, cov_f = m.predict(T(g[p + "xs"]), var="full")
            np.testing.assert_allclose(N(cov_f), g[p + "cov_full"], rtol=1e-6, atol=1e-10)
            np.testing.assert_allclose(N(mu_f), g[p + "mu_full"], rtol=1e-6, atol=1e-9)
            assert torch.equal(cov_f, cov_f.T)


def test_grbcm_full_covariance_batched_inversions_match_one_by_one(monkeypatch):
    """aggregate_full_covar (gr_bcm.py:99-114): the experts' m x m inversions as ONE batched call per step (default) against one
    expert after the other (PG_AGG_BATCH_MAX=0), m = 300 test points (padded to 512), and both against the oracle's aggregation."""
    from pygpr_amd import gr_bcm as _g

    rng = np.random.default_rng(91)
    nc, n, ng, d, m = 5, 200, 60, 3, 300
    xl, xg, xs = rng.random((nc, n, d)), rng.random((ng, d)), rng.random((m, d))
    yl, yg = np.sin(xl.sum(-1)), np.sin(xg.sum(-1))
    hp = np.concatenate([[1.0], np.full(d, 0.7), [0.05]])
    from pygpr_amd import gpr as _gpr

    outs = []
    for lim, vt_bytes in ((4096, _gpr._FULL_VT_BYTES), (0, 1), (4096, 3 * 512 * 512 * 8)):
        # second pass: every expert's products and update on their own; third: groups of three experts per launch
        monkeypatch.setattr(_g, "_AGG_BATCH_MAX", lim)
        monkeypatch.setattr(_gpr, "_FULL_VT_BYTES", vt_bytes)
        gm = pg.GRBCM(T(xl), T(yl), T(xg), T(yg), se_wn())
        gm.gpg.set_params(T(hp))
        gm.set_local_params(T(hp))
        mu, cov = gm.predict(T(xs), var="full")
        assert torch.equal(cov, cov.T)
        outs.append((N(mu), N(cov)))
    for other in outs[1:]:
        np.testing.assert_allclose(outs[0][0], other[0], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(outs[0][1], other[1], rtol=1e-8, atol=1e-12)
    mu_o, cov_o = orc.grbcm_predict([orc.SE, orc.WN], hp, np.tile(hp, (nc, 1)), xl, yl, xg, yg, xs, "full")
    np.testing.assert_allclose(outs[0][1], cov_o, rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(outs[0][0], mu_o, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("ng,nc,n,d", [(10, 2, 10, 2), (100, 5, 50, 3), (100, 10, 100, 7)])
def test_grbcm_reproduces_targets(ng, nc, n, d):
    """test_grbcm.py:18-37 restated: predicting one local shard reproduces sin(sum x)."""
    torch.manual_seed(ng + nc + n + d)
    xg = torch.rand(ng, d, dtype=torch.float64)
    xl = torch.rand(nc, n, d, dtype=torch.float64)
    m = pg.GRBCM(xl, torch.sin(xl.sum(-1)), xg, torch.sin(xg.sum(-1)), se_wn())
    mu, var = m.predict(xl[0], var="diag")
    assert torch.allclose(mu, torch.sin(xl[0].sum(-1)), atol=1e-4)


def test_grbcm_shared_hp_objective_matches_summed_oracle(golden):
    g = golden("grbcm")
    p = "g1_"
    m = pg.GRBCM(T(g[p + "xl"]), T(g[p + "yl"]), T(g[p + "xg"]), T(g[p + "yg"]), se_wn())
    hp = g[p + "hpg"].copy()
    loss, grad = pg.GRBCM_MLE(m).loss_and_grad(hp)
    x, y = orc.grbcm_data(g[p + "xl"], g[p + "yl"], g[p + "xg"], g[p + "yg"])
    ref = [orc.mle_loss_and_grad([orc.SE, orc.WN], hp, x[c], y[c], "kinv") for c in range(x.shape[0])]
    np.testing.assert_allclose(loss, sum(r[0] for r in ref), rtol=1e-10)
    gref = sum(r[1] for r in ref)
    np.testing.assert_allclose(grad, gref, rtol=1e-8, atol=1e-8 * np.abs(gref).max())


def test_grbcm_shared_hp_training_with_cg(golden):
    """the dead GRBCM.train is replaced by GRBCM_MLE + the stock CG driver (opt.py:45-67)."""
    g = golden("grbcm")
    p = "g0_"
    m = pg.GRBCM(T(g[p + "xl"]), T(g[p + "yl"]), T(g[p + "xg"]), T(g[p + "yg"]), se_wn())
    m.set_params(T(np.array([1.0, 1.0, 1.0, 1.0, 0.2])))
    loss = pg.GRBCM_MLE(m)
    f0 = float(loss.loss(N(m.params)))
    cg = pg.CG(loss)
    cg.args.update(maxiter=4, disp=False)
    cg.minimize()
    assert float(cg.res.fun) < f0 - 1.0                       # it descends
    assert torch.equal(m.gpg.params, torch.from_numpy(cg.res.x)) and m.gpl.params.shape == (2, 5)
    assert torch.equal(m.gpl.params[1], m.gpg.params) and m.need_upd
    mu, var = m.predict(T(g[p + "xs"]), var="diag")           # and the retrained committee predicts
    assert torch.isfinite(mu).all() and (var > 0).all()
    # same objective value as the oracle at the optimum found
    x, y = orc.grbcm_data(g[p + "xl"], g[p + "yl"], g[p + "xg"], g[p + "yg"])
    ref = sum(orc.mle_loss(( [orc.SE, orc.WN] ), cg.res.x, x[c], y[c]) for c in range(x.shape[0]))
    np.testing.assert_allclose(float(loss.loss(cg.res.x)), ref, rtol=1e-9)


# ------------------------------------------------------------------ fp32 opt-in and full-size properties
def test_fp32_opt_in():
    n, d = 600, 8
    x, y = orc.synth(n, d, seed=2)
    xp = np.random.default_rng(1).random((40, d))
    hp = np.concatenate([[1.0], np.full(d, 0.7), [0.1]])
    cov = pg.Compose([pg.Matern52(), pg.White_noise()])
    gp = pg.Exact_GP(T(x).float(), T(y).float(), cov)
    gp.set_params(T(hp))
    mu, var = gp.predict(T(xp).float(), var="diag")
    assert mu.dtype == torch.float32
    mu_ref, var_ref = orc.gp_predict([orc.M52, orc.WN], hp, x, y, xp, "diag")
    np.testing.assert_allclose(N(mu), mu_ref, atol=2e-3)
    np.testing.assert_allclose(N(var), var_ref, atol=2e-3)
    mu_f, cov_f = gp.predict(T(xp).float(), var="full")        # the test-point-major products in fp32 (gpr.py:108-120)
    _, cov_ref = orc.gp_predict([orc.M52, orc.WN], hp, x, y, xp, "full")
    assert cov_f.dtype == torch.float32 and torch.equal(cov_f, cov_f.T)
    np.testing.assert_allclose(N(cov_f), cov_ref, atol=2e-3)
    np.testing.assert_allclose(N(mu_f), mu_ref, atol=2e-3)
    loss, grad = pg.MLE(gp).loss_and_grad(hp.copy())
    l_ref, g_ref = orc.mle_loss_and_grad([orc.M52, orc.WN], hp, x, y, "kinv")
    np.testing.assert_allclose(loss, l_ref, rtol=1e-3)
    np.testing.assert_allclose(grad, g_ref, rtol=2e-2, atol=2e-2 * np.abs(g_ref).max())


def test_grbcm_fp32_committee(golden):
    """BASELINE config 5's layout at fixture size: fp32 committee (Matern-5/2 + noise, shared hp) against the fp64
    oracle -- rtol 1e-3 class."""
    g = golden("grbcm")
    p = "g1_"
    cov = pg.Compose([pg.Matern52(), pg.White_noise()])
    f32 = lambda a: T(a).float()
    m = pg.GRBCM(f32(g[p + "xl"]), f32(g[p + "yl"]), f32(g[p + "xg"]), f32(g[p + "yg"]), cov)
    hp = np.array([1.0, 0.8, 0.8, 0.8, 0.1])
    m.set_params(T(hp))
    mu, var = m.predict(f32(g[p + "xs"]), var="diag")
    assert mu.dtype == torch.float32
    nc = g[p + "xl"].shape[0]
    mu_ref, var_ref = orc.grbcm_predict([orc.M52, orc.WN], hp, np.broadcast_to(hp, (nc, hp.size)), g[p + "xl"], g[p + "yl"],
                                        g[p + "xg"], g[p + "yg"], g[p + "xs"])[:2]
    np.testing.assert_allclose(N(mu), mu_ref, atol=2e-3)
    np.testing.assert_allclose(N(var), var_ref, rtol=2e-2, atol=1e-4)
    # the full-covariance committee in fp32: batched products and batched m x m inversions (gr_bcm.py:99-114)
    mu_f, cov_f = m.predict(f32(g[p + "xs"]), var="full")
    mu_fr, cov_fr = orc.grbcm_predict([orc.M52, orc.WN], hp, np.broadcast_to(hp, (nc, hp.size)), g[p + "xl"], g[p + "yl"],
                                      g[p + "xg"], g[p + "yg"], g[p + "xs"], "full")
    assert cov_f.dtype == torch.float32 and torch.equal(cov_f, cov_f.T)
    np.testing.assert_allclose(N(cov_f), cov_fr, rtol=5e-2, atol=5e-4)
    np.testing.assert_allclose(N(mu_f), mu_fr, rtol=5e-2, atol=5e-3)
    loss, grad = pg.GRBCM_MLE(m).loss_and_grad(hp.copy())
    x, y = orc.grbcm_data(g[p + "xl"], g[p + "yl"], g[p + "xg"], g[p + "yg"])
    ref = [orc.mle_loss_and_grad([orc.M52, orc.WN], hp, x[c], y[c], "kinv") for c in range(nc)]
    np.testing.assert_allclose(loss, sum(r[0] for r in ref), rtol=1e-3)
    gref = sum(r[1] for r in ref)
    np.testing.assert_allclose(grad, gref, rtol=2e-2, atol=2e-2 * np.abs(gref).max())


def test_fp32_tracks_fp64_at_scale():
    """BASELINE config 5's regime at a size with 32 outer panels: the fp32 path against the fp64 HIP path (itself
    pinned against the oracle above) on the same data -- rtol 1e-3 class, sigma_n = 0.1."""
    n, d = 8192, 16
    x, y = orc.synth(n, d, seed=77)
    hp = np.concatenate([[1.0], np.full(d, 0.5), [0.1]])
    cov = pg.Compose([pg.Matern52(), pg.White_noise()])
    l64, g64 = pg.MLE(pg.Exact_GP(T(x), T(y), cov)).loss_and_grad(hp.copy())
    l32, g32 = pg.MLE(pg.Exact_GP(T(x).float(), T(y).float(), cov)).loss_and_grad(hp.copy())
    np.testing.assert_allclose(l32, l64, rtol=1e-3)
    np.testing.assert_allclose(g32, g64, rtol=2e-2, atol=2e-2 * np.abs(g64).max())


def test_cfg2_size_properties():
    """BASELINE config 2 (N=8192, D=8, fp64): size-independent properties instead of an O(n^3) oracle run --
    (i) the gradient agrees with central differences of the loss along a random direction,
    (ii) predicting the training inputs gives K alpha' = y - (sigma_n^2 + jitter) alpha, i.e. the factor solves K."""
    n, d = 8192, 8
    x, y = orc.synth(n, d, seed=1234)
    hp = np.concatenate([[1.0], np.ones(d), [0.1]])
    gp = pg.Exact_GP(T(x), T(y), se_wn())
    mle = pg.MLE(gp)
    loss, grad = mle.loss_and_grad(hp.copy())
    v = np.random.default_rng(0).standard_normal(hp.size)
    v /= np.linalg.norm(v)
    eps = 1e-5
    fd = (mle.loss(hp + eps * v) - mle.loss(hp - eps * v)) / (2 * eps)
    np.testing.assert_allclose(grad @ v, fd, rtol=1e-5)
    gp.set_params(T(hp))
    mu, _ = gp.predict(T(x[:2048]), var="none")
    alpha = N(gp.wt)
    np.testing.assert_allclose(N(mu), y[:2048] - (hp[-1] ** 2 + 1e-7) * alpha[:2048], atol=1e-8)


@pytest.mark.parametrize("case", __import__("surface_cases").ALL, ids=lambda f: f.__name__)
def test_surface_round2(golden, case):
    """distance(), an 11-child Compose, batched test points, batched params on shared points, batch-of-one squeezes, the
    rank-one covariance the reference factorises, memo invalidation -- against fixtures captured from the reference."""
    case(golden("surface2"))


def test_large_d_through_the_class_surface():
    """d = 40 and d = 64 (PG_MAX_DIM): cov.kernel (mirrored build), predict(var="full") (K** mirrored build) and the
    NLML gradient against the oracle."""
    for d in (40, 64):
        n, m = 300, 33
        rng = np.random.default_rng(d)
        x, xp = rng.random((n, d)), rng.random((m, d))
        y = np.sin(-x.sum(1)) + 0.1 * rng.standard_normal(n)
        hp = np.concatenate([[1.0], (0.3 + 0.3 * rng.random(d)) / np.sqrt(d), [0.1]])
        covs = [orc.SE, orc.WN]
        np.testing.assert_allclose(N(se_wn().kernel(T(hp), T(x))), orc.kernel(covs, hp, x), atol=1e-13)
        gp = pg.Exact_GP(T(x), T(y), se_wn())
        gp.set_params(T(hp))
        mu, cov = gp.predict(T(xp), var="full")
        mu_r, cov_r = orc.gp_predict(covs, hp, x, y, xp, "full")
        np.testing.assert_allclose(N(mu), mu_r, atol=1e-9)
        np.testing.assert_allclose(N(cov), cov_r, atol=1e-10)
        loss, grad = pg.MLE(gp).loss_and_grad(hp.copy())
        l_r, g_r = orc.mle_loss_and_grad(covs, hp, x, y, "kinv")
        np.testing.assert_allclose(loss, l_r, rtol=1e-10)
        np.testing.assert_allclose(grad, g_r, rtol=1e-8, atol=1e-8 * np.abs(g_r).max())


def test_headline_size_properties():
    """BASELINE config 3's size, N=16384 D=8 fp64 -- the schedule bench.py times (1024-column outer panels, the last 8192 rows
    on the flag-coupled chain): (i) directional derivative against central differences, (ii) K alpha = y through a
    prediction at training inputs, (iii) the fused pg_potrf_trtri factor equals the separate pg_potrf (to rounding: the fused call
    takes the recursive split at this size) and its inverse undoes it."""
    from pygpr_amd._ops import get_ops

    n, d = 16384, 8
    x, y = orc.synth(n, d, seed=1234)
    hp = np.concatenate([[1.0], np.ones(d), [0.1]])
    gp = pg.Exact_GP(T(x), T(y), se_wn())
    mle = pg.MLE(gp)
    mle.memoize = False
    loss, grad = mle.loss_and_grad(hp.copy())
    assert np.isfinite(loss) and np.isfinite(grad).all()
    v = np.random.default_rng(0).standard_normal(hp.size)
    v /= np.linalg.norm(v)
    eps = 1e-5
    fd = (mle.loss(hp + eps * v) - mle.loss(hp - eps * v)) / (2 * eps)
    np.testing.assert_allclose(grad @ v, fd, rtol=2e-5)
    gp.set_params(T(hp))
    mu, _ = gp.predict(T(x[:1024]), var="none")
    alpha = N(gp.wt)
    np.testing.assert_allclose(N(mu), y[:1024] - (hp[-1] ** 2 + 1e-7) * alpha[:1024], atol=2e-8)
    del gp, mle
    torch.cuda.empty_cache()
    ops = get_ops()
    spec, _ = pg.covar.spec_of(se_wn(), d)
    xd, hpd = T(x).cuda(), T(hp).cuda()
    a, b = ops.empty(n, n), ops.empty(n, n)
    ops.kernel_build(spec, hpd, xd, None, a, lower_only=True, jitter=1e-7)
    b.copy_(a)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    invd, invd2 = ops.potrf_workspace(n, torch.float64), ops.potrf_workspace(n, torch.float64)
    minv = ops.empty(n, n)
    ops.potrf_trtri(a, invd, info, minv)
    assert int(info.item()) == 0
    ops.potrf(b, invd2, info)
    # Round 4: from n = 16384 the fused call splits the matrix at n / 2 and takes everything that crosses the split as four large
    # products (linalg.hip: potrf_trtri_rec), so its factor equals pg_potrf's to rounding, not bit for bit (measured 2e-13 on
    # entries of size <= 1.1); PG_REC_MIN=0 restores the one-level schedule and the bitwise equality.
    fa, fb = torch.tril(a), torch.tril(b)
    if os.environ.get("PG_REC_MIN") == "0":
        assert torch.equal(fa, fb)
    else:
        err = float((fa - fb).abs().max())
        print("fused (recursive split) vs plain factor at n = 16384: max abs difference %.2e" % err)
        assert err <= 5e-12
    del b, fa, fb
    g = torch.Generator(device="cuda").manual_seed(5)
    w = torch.randn(n, device="cuda", dtype=torch.float64, generator=g)
    low = torch.tril(a)
    assert float((torch.tril(minv) @ (low @ w) - w).abs().max()) <= 1e-8


def test_cfg5_size_fp32_tracks_fp64():
    """BASELINE config 5's per-expert size, n = 33792 (two levels of the recursive split, 2048-column outer panels below it), Matern-5/2, D = 16: the fp32
    NLML against the fp64 HIP path on the same data, rtol 1e-3 (sigma_n = 0.1)."""
    n, d = 33792, 16
    x, y = orc.synth(n, d, seed=55)
    hp = np.concatenate([[1.0], np.full(d, 0.5), [0.1]])
    cov = pg.Compose([pg.Matern52(), pg.White_noise()])
    m64 = pg.MLE(pg.Exact_GP(T(x), T(y), cov))
    l64, g64 = m64.loss_and_grad(hp.copy())
    del m64
    torch.cuda.empty_cache()
    l32, g32 = pg.MLE(pg.Exact_GP(T(x).float(), T(y).float(), cov)).loss_and_grad(hp.copy())
    np.testing.assert_allclose(l32, l64, rtol=1e-3)
    np.testing.assert_allclose(g32, g64, rtol=5e-2, atol=5e-2 * np.abs(g64).max())


def test_cfg5_size_nlml_against_the_oracle():
    """BASELINE config 5's per-expert size, n = 33792, Matern-5/2, D = 16, against the ORACLE itself (round 5; the test above can only compare
    the fp32 HIP path with the fp64 HIP path): the oracle's NLML at the full size (`matern52_nlml_lean`: direct differences in row slabs, LAPACK factorisation on
    the host's threads, 20 GB) -- against the fp64 HIP path (measured 5.5e-14, asserted 1e-10: cond(K) is a few 1e4 at
    sigma_n = 0.1) and the fp32 HIP path (measured 4.7e-6, asserted 1e-4; SURVEY 8c's fp32 class is 1e-3).  About 50 s, most of it the host's.  The gradient at this size stays with the HIP-vs-HIP
    comparison above and the oracle comparison at n = 4096 below (eighteen n x n derivative matrices on the host are not a test)."""
    n, d = 33792, 16
    x, y = orc.synth(n, d, seed=55)
    hp = np.concatenate([[1.0], np.full(d, 0.5), [0.1]])
    l_ref = orc.matern52_nlml_lean(hp, x, y)
    cov = pg.Compose([pg.Matern52(), pg.White_noise()])
    m64 = pg.MLE(pg.Exact_GP(T(x), T(y), cov))
    l64 = float(m64.loss(hp.copy()))
    del m64
    torch.cuda.empty_cache()
    l32 = float(pg.MLE(pg.Exact_GP(T(x).float(), T(y).float(), cov)).loss(hp.copy()))
    print("\n[cfg5 size, n = 33792] NLML oracle %.6f  fp64 HIP rel err %.2e  fp32 HIP rel err %.2e" % (
        l_ref, abs(l64 - l_ref) / abs(l_ref), abs(l32 - l_ref) / abs(l_ref)))
    np.testing.assert_allclose(l64, l_ref, rtol=1e-10)
    np.testing.assert_allclose(l32, l_ref, rtol=1e-4)


def test_cfg5_kernel_midsize_against_the_oracle():
    """BASELINE config 5's covariance (Matern-5/2 + noise, D = 16) at n = 4096 -- the largest size the CPU oracle evaluates in seconds --
    in fp64 AND fp32 against the ORACLE (not against each other, as the full-size test above has to): NLML and gradient of the fp64 HIP
    path to 1e-10 / 1e-8, of the fp32 path to the fp32 class (measured 3.9e-5 / 1.3e-4; asserted 1e-3 / 2e-3 of |g|inf).  The Matern kernel itself has no reference: the oracle is pinned to sklearn's Matern(nu=2.5) and finite
    differences in tests/test_oracle_golden.py."""
    n, d = 4096, 16
    x, y = orc.synth(n, d, seed=56)
    hp = np.concatenate([[1.0], np.full(d, 0.5), [0.1]])
    covs = [orc.M52, orc.WN]
    l_ref, g_ref = orc.mle_loss_and_grad(covs, hp, x, y, "kinv")
    cov = pg.Compose([pg.Matern52(), pg.White_noise()])
    l64, g64 = pg.MLE(pg.Exact_GP(T(x), T(y), cov)).loss_and_grad(hp.copy())
    np.testing.assert_allclose(l64, l_ref, rtol=1e-10)
    np.testing.assert_allclose(g64, g_ref, rtol=1e-8, atol=1e-8 * np.abs(g_ref).max())
    l32, g32 = pg.MLE(pg.Exact_GP(T(x).float(), T(y).float(), cov)).loss_and_grad(hp.copy())
    print("\n[cfg5 kernel, n = 4096] fp32 vs oracle: NLML rel err %.2e, grad err/|g|inf %.2e" % (
        abs(l32 - l_ref) / abs(l_ref), np.abs(g32 - g_ref).max() / np.abs(g_ref).max()))
    np.testing.assert_allclose(l32, l_ref, rtol=1e-3)
    assert np.abs(g32 - g_ref).max() <= 2e-3 * np.abs(g_ref).max()


def _check_sampler(golden):
    g = golden("sampler")
    mins, maxs = T(g["mins"]), T(g["maxs"])
    assert torch.equal(pg.UNIFORM(3).sample(50, mins, maxs), T(g["uni"]))          # same generator call sequence
    m1 = pg.MATERN1(5)
    assert torch.equal(m1.sample(12, mins, maxs), T(g["mat"]))
    np.testing.assert_allclose(float(m1.min_dist), g["mat_min_dist"], rtol=1e-14)
    xpart, xc = pg.MATERN1(7).partition(4, 25, mins, maxs)
    assert torch.equal(xc, T(g["part_xc"])) and torch.equal(xpart, T(g["part_x"]))
    np.testing.assert_allclose(pg.euclidean_dist(T(g["ed_x"]), T(g["ed_y"])).numpy(), g["ed"], atol=1e-14)
    assert torch.equal(pg.cluster_samples(T(g["cs_x"]), T(g["part_xc"])), T(g["cs"]))
    # the shards feed GRBCM directly (this is what the reference's test_grbcm.py does with MATERN1.partition)
    m = pg.GRBCM(xpart, torch.sin(xpart.sum(-1)), xc, torch.sin(xc.sum(-1)), pg.Compose([pg.Squared_exponential(), pg.White_noise()]))
    assert (m.nc, m.nsc, m.ng, m.dim) == (4, 25, 4, 2)


def test_samplers_and_partition(golden):
    """the nearest-centre assignment / distance matrix of sampler.py on the device."""
    _check_sampler(golden)
    rng = np.random.default_rng(0)
    x, c = rng.random((5000, 7)), rng.random((300, 7))
    from pygpr_amd.sampler import nearest_centre
    d2 = ((x[:, None, :] - c[None, :, :]) ** 2).sum(2)
    assert np.array_equal(N(nearest_centre(T(x), T(c))), np.argmin(d2, axis=1))
    np.testing.assert_allclose(N(pg.euclidean_dist(T(x), T(c))), d2, atol=1e-13)


_NCCL_SCRIPT = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1", PG_DIST_SINGLE_RANK="1",
                  HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import pygpr_amd as pg
g = np.load(os.path.join(sys.argv[1], "tests", "golden", "grbcm.npz"))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
p = "g1_"
cov = pg.Compose([pg.Squared_exponential(), pg.White_noise()])
m = pg.GRBCM(T(g[p + "xl"]), T(g[p + "yl"]), T(g[p + "xg"]), T(g[p + "yg"]), cov, distributed=True)
assert m.distributed and dist.get_backend() == "nccl"
m.gpg.set_params(T(g[p + "hpg"])); m.set_local_params(T(g[p + "hpl"]))
mu, var = m.predict(T(g[p + "xs"]), var="diag")
np.testing.assert_allclose(mu.numpy(), g[p + "mu"], atol=1e-10)
np.testing.assert_allclose(var.numpy(), g[p + "var"], atol=1e-11)
m.set_params(T(g[p + "hpg"]))
mu_f, cov_f = m.predict(T(g[p + "xs"]), var="full")
np.testing.assert_allclose(cov_f.numpy(), g[p + "cov_full"], rtol=1e-6, atol=1e-10)
loss, grad = pg.GRBCM_MLE(m).loss_and_grad(g[p + "hpg"].copy())
ref = pg.GRBCM(T(g[p + "xl"]), T(g[p + "yl"]), T(g[p + "xg"]), T(g[p + "yg"]), cov)
ref.set_params(T(g[p + "hpg"]))
l2, g2 = pg.GRBCM_MLE(ref).loss_and_grad(g[p + "hpg"].copy())
assert float(loss) == float(l2) and np.array_equal(grad, g2)
xl = g[p + "xl"].copy(); xl[-1, 3, 0] = np.nan
bad = pg.GRBCM(T(xl), T(g[p + "yl"]), T(g[p + "xg"]), T(g[p + "yg"]), cov, distributed=True)
bad.set_params(T(g[p + "hpg"]))
for call in (lambda: bad.predict(T(g[p + "xs"]), var="diag"), lambda: pg.GRBCM_MLE(bad).loss_and_grad(g[p + "hpg"].copy())):
    try:
        call(); raise SystemExit("did not raise")
    except torch.linalg.LinAlgError:
        pass
dist.barrier(); dist.destroy_process_group()
print("NCCL-OK")
"""


def test_rccl_code_path_single_rank(tmp_path):
    """The `nccl` (= RCCL) branches of GRBCM / GRBCM_MLE -- device-side all-reduces of the [3, m] + status buffer, the
    [m_pad, m_pad] weighted precision, the [1 + nhp] + status vector, the status relay of a non-PD expert -- run end to end
    on this one-GPU box over a one-rank RCCL communicator (a fresh process: the process group is global state)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "nccl_one_rank.py"
    script.write_text(_NCCL_SCRIPT)
    r = subprocess.run([sys.executable, str(script), root, str(29500 + os.getpid() % 2000)], capture_output=True, text=True,
                       timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0 and "NCCL-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.gpu
def test_class_surface_survives_a_coupled_chain_timeout():
    """Exact_GP.update / predict, MLE.loss_and_grad and the full-covariance committee with every wait of the coupled chain forced
    to expire: each repeats its evaluation on the classic chain and returns the oracle's numbers -- no exception, as
    tc.cholesky (PyGPR/gpr.py:69) never fails on a positive-definite matrix."""
    from pygpr_amd._ops import get_ops

    ops = get_ops()
    n, d, m = 1700, 4, 40          # pads to 1792: four outer panels, all on the coupled chain
    x, y = orc.synth(n, d, seed=3)
    xp = np.random.default_rng(8).random((m, d))
    hp = np.concatenate([[1.1], np.full(d, 0.9), [0.1]])
    covs = [orc.SE, orc.WN]
    l_ref, g_ref = orc.mle_loss_and_grad(covs, hp, x, y, "kinv")
    mu_ref, var_ref = orc.gp_predict(covs, hp, x, y, xp, "diag")
    assert ops.coupled_chain() == 1
    try:
        for eager in (False, True):
            ops.set_coupled_chain(1)
            ops.set_spin_budget(-1)
            before = ops.chain_timeouts()
            gp = pg.Exact_GP(T(x), T(y), pg.Compose([pg.Squared_exponential(), pg.White_noise()]), eager_inverse=eager)
            gp.set_params(T(hp))
            mu, var = gp.predict(T(xp), var="diag")
            assert ops.chain_timeouts() == before + 1 and ops.coupled_chain() == 0
            np.testing.assert_allclose(N(mu), mu_ref, atol=1e-9)
            np.testing.assert_allclose(N(var), var_ref, atol=1e-10)
        ops.set_coupled_chain(1)
        ops.set_spin_budget(-1)
        before = ops.chain_timeouts()
        loss = pg.MLE(gp)
        loss.memoize = False
        val, grad = loss.loss_and_grad(hp.copy())
        assert ops.chain_timeouts() == before + 1 and ops.coupled_chain() == 0
        np.testing.assert_allclose(val, l_ref, rtol=1e-10)
        np.testing.assert_allclose(grad, g_ref, rtol=1e-8, atol=1e-8 * np.abs(g_ref).max())
    finally:
        ops.set_spin_budget(0)
        ops.set_coupled_chain(1)
    assert ops.coupled_chain() == 1
    val2, _ = pg.MLE(gp).loss_and_grad(hp.copy())          # re-armed: the coupled chain runs again and agrees
    assert ops.last_coupled_panels() > 0
    np.testing.assert_allclose(val2, l_ref, rtol=1e-10)


@pytest.mark.gpu
def test_cfg4_size_committee():
    """BASELINE config 4 at its real size: 8 experts x (1024 global + 8192 own) points, D = 16, one batch of 8192 test points.
    (i) the device aggregation (pg_grbcm_local_terms / pg_grbcm_finish, PyGPR/gr_bcm.py:116-149) against the oracle's GRBCM.aggregate
    fed with the same per-expert device means / variances; (ii) every expert's factor solves its system, K alpha = y at 1024 of its
    own training points (PyGPR/gpr.py:65-85); (iii) positive variances below the prior's.  bench.py's grbcm_predict leg asserts the
    same (bench.check_committee)."""
    from bench import check_committee, synth_expert

    nc, nls, ng, d, m = 8, 8192, 1024, 16, 8192
    xg, yg = synth_expert(ng, d, 7)
    sh = [synth_expert(nls, d, 100 + c) for c in range(nc)]
    g4 = pg.GRBCM(T(np.stack([s[0] for s in sh])), T(np.stack([s[1] for s in sh])), T(xg), T(yg), se_wn())
    hp = np.concatenate([[1.0], np.full(d, 0.5), [0.1]])
    g4.set_params(T(hp))
    xs = T(np.random.default_rng(4321).random((m, d)))
    rep = check_committee(g4, xs, hp)
    assert rep["experts_checked"] == nc and rep["batch"] == m
    mu, var = g4.predict(xs, var="diag")
    assert float(var.min()) > 0 and float(var.max()) < 1.0 + 0.01 + 1e-6      # prior variance sigma^2 + sigma_n^2
    assert float((mu - T(np.sin(-N(xs).sum(1)))).abs().mean()) < 0.2            # and it predicts the function it was given
    # (iv) the FULL-covariance committee at this size (gr_bcm.py:99-114,151-155; m = 2048): every expert's Vt and rank-n update in one
    # launch and nine 2048 x 2048 inversions on the batched coupled chain, against one expert / one inversion after the other;
    # exactly symmetric and positive definite.  (Its MEAN is not compared with the diag committee's: the reference scales the summed
    # precision-weighted means by diag(cov_full) -- gr_bcm.py:147 -- which is a different estimator; the golden vectors pin that formula.)
    from pygpr_amd import gpr as _gpr, gr_bcm as _g

    xf = xs[:2048]
    mu_f, cov_f = g4.predict(xf, var="full")
    keep = (_g._AGG_BATCH_MAX, _gpr._FULL_VT_BYTES)
    try:
        _g._AGG_BATCH_MAX, _gpr._FULL_VT_BYTES = 0, 1
        mu_1, cov_1 = g4.predict(xf, var="full")
    finally:
        _g._AGG_BATCH_MAX, _gpr._FULL_VT_BYTES = keep
    assert torch.equal(cov_f, cov_f.T) and bool(torch.isfinite(cov_f).all())
    scale = float(cov_1.diagonal().mean())
    assert float((cov_f - cov_1).abs().max()) < 1e-9 * scale and float((mu_f - mu_1).abs().max()) < 1e-9
    torch.linalg.cholesky(cov_f.cuda())                                          # positive definite (raises otherwise)
    assert float(cov_f.diagonal().min()) > 0
    del g4
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_default_hp_gradient_error_is_stated(golden, capsys):
    """Default hyper-parameters (sigma_n = 1e-4 + jitter 1e-7: cond(K) ~ 1e9): SURVEY 8c states NLML 1e-8 / gradient 1e-6 under the
    rule |delta| <= 50 eps cond(K) |ref|.  The measured errors against the reference's own numbers are printed and asserted against
    BOTH the rule (with cond(K) computed here) and the fixed 1e-6."""
    g = golden("gp")
    covs = [orc.SE, orc.WN]
    rows = []
    for tag in ("a", "b"):
        x, y, hp0 = g[tag + "_x"], g[tag + "_y"], g[tag + "_hp0"]
        k = orc.kernel(covs, hp0, x, form="direct") + 1e-7 * np.eye(x.shape[0])
        cond = np.linalg.cond(k)
        l0, g0 = pg.MLE(pg.Exact_GP(T(x), T(y), se_wn())).loss_and_grad(hp0.copy())
        lref = g[tag + "_loss0"] if tag == "a" else np.array(-3461.42170686)
        gref = g[tag + "_grad0"]
        el = abs(float(l0) - float(lref)) / abs(float(lref))
        eg = float(np.abs(g0 - gref).max() / np.abs(gref).max())
        rule = 50 * np.finfo(np.float64).eps * cond
        rows.append((tag, x.shape[0], cond, el, eg, rule))
        assert el < 1e-8 and el <= rule, rows
        assert eg <= rule, rows
        assert eg < 1e-7, rows          # (SURVEY 8c's figure for this class is 1e-6)
    with capsys.disabled():
        for r in rows:
            print("\n[default-hp parity] case %s n=%d cond(K)=%.2e  NLML rel err %.2e  grad err/|g|inf %.2e  (rule 50 eps cond = %.1e)" % r)


@pytest.mark.gpu
def test_reference_test_sizes_batched(golden):
    """The reference's own test sizes (PyGPR/tests/test_gpr.py:59-100: nc = 10 experts of n = 100 points; tests/test_grbcm.py:18-37:
    nc = 5, ng = 100, nls = 50) against what the imported reference returned (round3.npz).  The experts of such a model are
    factorised in one batched call (every launch covers all ten)."""
    from pygpr_amd._ops import get_ops

    r = golden("round3")
    gp = pg.Exact_GP(T(r["sm_x"]), T(r["sm_y"]), se_wn())
    gp.set_params(T(r["sm_hp"]))
    mu, var = gp.predict(T(r["sm_xp"]), var="diag")
    assert gp._bat is not None and get_ops().last_coupled_panels() == 0          # the batched path ran
    np.testing.assert_allclose(N(mu), r["sm_mu"], atol=1e-10)
    np.testing.assert_allclose(N(var), r["sm_var"], atol=1e-11)
    np.testing.assert_allclose(N(gp.wt), r["sm_wt"], rtol=1e-9, atol=1e-9)
    mle = pg.MLE(gp)
    loss, grad = mle.loss_and_grad(r["sm_hp"].copy())
    assert mle.last_batched                              # round 4: the gradient path ran experts-together too (one sync, ~20 launches)
    np.testing.assert_allclose(loss, r["sm_loss"], rtol=1e-10)
    np.testing.assert_allclose(grad, r["sm_grad"], rtol=1e-8, atol=1e-8 * np.abs(r["sm_grad"]).max())
    mle.memoize = False
    os.environ["PG_MLE_SERIAL"] = "1"                    # the per-expert loop of rounds 1-3: same numbers to rounding
    try:
        loss1, grad1 = mle.loss_and_grad(r["sm_hp"].copy())
    finally:
        del os.environ["PG_MLE_SERIAL"]
    assert not mle.last_batched
    np.testing.assert_allclose(loss, loss1, rtol=1e-12)
    np.testing.assert_allclose(grad, grad1, rtol=1e-9, atol=1e-10 * np.abs(grad1).max())
    lo = mle.loss(r["sm_hp"].copy())                     # loss only, batched as well (n <= 4096: through the inverse factors)
    assert mle.last_batched
    np.testing.assert_allclose(lo, r["sm_loss"], rtol=1e-10)
    nc = r["sg_xl"].shape[0]
    g = pg.GRBCM(T(r["sg_xl"]), T(r["sg_yl"]), T(r["sg_xg"]), T(r["sg_yg"]), se_wn())
    g.set_params(T(r["sg_hp"]))
    mu, var = g.predict(T(r["sg_xl"][2]), var="diag")
    assert g.gpl._bat is not None and len(g.gpl._experts) == nc
    np.testing.assert_allclose(N(mu), r["sg_mu"], atol=1e-9)
    np.testing.assert_allclose(N(var), r["sg_var"], rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(N(g.beta), r["sg_beta"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(mu.numpy(), np.sin(r["sg_xl"][2].sum(-1)), atol=1e-2)      # test_grbcm.py:36 (its data has no noise)


@pytest.mark.gpu
def test_batched_diag_prediction_matches_the_per_expert_loop(golden, monkeypatch):
    """Exact_GP.predict / GRBCM.predict(var="diag") of a batched model (round 5): every expert's K* in one launch, means and variances in
    three, the committee's terms in one -- against the per-expert loop of rounds 1-4 (PG_PREDICT_SERIAL=1), BIT FOR BIT, and against the
    oracle.  Cases: the reference's own test size (ten experts of 100 points, tests/test_gpr.py:59-100), ragged sizes with shared and
    per-expert test points, shared training points under batched params, more test points than one chunk, a lazily inverted model."""
    from pygpr_amd import gpr as _gpr

    rng = np.random.default_rng(77)

    def both(fn):
        out_b = fn()
        monkeypatch.setenv("PG_PREDICT_SERIAL", "1")
        try:
            out_s = fn()
        finally:
            monkeypatch.delenv("PG_PREDICT_SERIAL")
        return out_b, out_s

    for nc, n, d, m, per_expert_xp in ((10, 100, 3, 100, False), (3, 300, 5, 77, True), (4, 700, 16, 300, False)):
        x = rng.random((nc, n, d))
        y = np.sin(-x.sum(-1)) + 0.1 * rng.standard_normal((nc, n))
        hp = np.stack([np.concatenate([[1.0 + 0.05 * c], 0.6 + 0.3 * rng.random(d), [0.1]]) for c in range(nc)])
        xp = rng.random((nc, m, d)) if per_expert_xp else rng.random((m, d))
        gp = pg.Exact_GP(T(x), T(y), se_wn(), eager_inverse=(nc != 3))
        gp.set_params(T(hp))
        (mu_n, _), (mu_ns, _) = both(lambda: gp.predict(T(xp), var="none"))
        (mu_b, v_b), (mu_s, v_s) = both(lambda: gp.predict(T(xp), var="diag"))
        assert gp._bat is not None
        assert torch.equal(mu_b, mu_s) and torch.equal(v_b, v_s) and torch.equal(mu_n, mu_ns) and torch.equal(mu_n, mu_b)
        gp.predict(T(xp), var="diag")
        assert gp.last_predict_batched
        for c in range(nc):
            mu_o, var_o = orc.gp_predict([orc.SE, orc.WN], hp[c], x[c], y[c], xp[c] if per_expert_xp else xp, "diag", form="direct")
            np.testing.assert_allclose(N(mu_b[c]), mu_o, atol=1e-9)
            np.testing.assert_allclose(N(v_b[c]), var_o, atol=1e-10)
    # batched params on shared training points (x [n, d], params [nc, nhp]) and more test points than one chunk
    monkeypatch.setattr(_gpr, "_CHUNK", 256)
    n, d, m, nc = 200, 4, 700, 3
    x = rng.random((n, d)); y = np.sin(-x.sum(-1)) + 0.1 * rng.standard_normal(n)
    hp = np.stack([np.concatenate([[1.0], 0.5 + 0.2 * c + rng.random(d), [0.1]]) for c in range(nc)])
    gp = pg.Exact_GP(T(x), T(y), se_wn())
    gp.set_params(T(hp))
    xp = rng.random((m, d))
    (mu_b, v_b), (mu_s, v_s) = both(lambda: gp.predict(T(xp), var="diag"))
    assert gp.last_predict_batched is False and mu_b.shape == (nc, m)      # (the serial run came last)
    assert torch.equal(mu_b, mu_s) and torch.equal(v_b, v_s)
    for c in range(nc):
        mu_o, var_o = orc.gp_predict([orc.SE, orc.WN], hp[c], x, y, xp, "diag", form="direct")
        np.testing.assert_allclose(N(mu_b[c]), mu_o, atol=1e-9)
        np.testing.assert_allclose(N(v_b[c]), var_o, atol=1e-10)
    # the committee: batched experts + one launch for all local terms, against the loop and the golden committee
    g = golden("grbcm")
    p = "g1_"
    com = pg.GRBCM(T(g[p + "xl"]), T(g[p + "yl"]), T(g[p + "xg"]), T(g[p + "yg"]), se_wn())
    com.gpg.set_params(T(g[p + "hpg"]))
    com.gpl.set_params(T(g[p + "hpl"]))

    def committee():
        mu, var = com.predict(T(g[p + "xs"]), var="diag")
        return mu, var, com.beta.clone(), com.prec.clone()

    (mu_b, v_b, be_b, pr_b), (mu_s, v_s, be_s, pr_s) = both(committee)
    assert torch.equal(mu_b, mu_s) and torch.equal(v_b, v_s) and torch.equal(be_b, be_s) and torch.equal(pr_b, pr_s)
    np.testing.assert_allclose(N(mu_b), g[p + "mu"], atol=1e-10)
    np.testing.assert_allclose(N(v_b), g[p + "var"], atol=1e-11)
    np.testing.assert_allclose(N(be_b), g[p + "beta"], atol=1e-9)
