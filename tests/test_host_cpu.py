"""CPU tier (-m "not gpu"): the C-ABI library loads and exports every symbol include/pygpr_hip.h declares (no
compute calls), the product refuses to run without a GPU, and the HOST logic of the PyGPR-compatible classes
is checked over a test double of the device-op layer (tests/oracle_ops.py) against the golden vectors."""
import ctypes
import os

import numpy as np
import pytest
import torch

import pygpr_amd as pg
from pygpr_amd import _lib, _ops
from oracle import pygpr_oracle as orc
from oracle_ops import OracleOps


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def se_wn():
    return pg.Compose([pg.Squared_exponential(), pg.White_noise()])


@pytest.fixture
def fake_ops(monkeypatch, tmp_path):
    monkeypatch.setattr(_ops, "_OPS", OracleOps())
    monkeypatch.chdir(tmp_path)


def test_library_loads_and_exports_every_header_symbol():
    lib = _lib.load(check_symbols=True)
    names = _lib.header_symbols()
    assert len(names) >= 24 and "pg_potrf" in names and "pg_kernel_build" in names
    for n in names:
        assert isinstance(getattr(lib, n), ctypes._CFuncPtr)
    assert lib.pg_version() == 100
    # inv_diag + the coupled chain's flag words (the panel-mode buffers only exist with PG_PANEL_MODE=1)
    assert lib.pg_potrf_worksize(0, 512) == 512 * 128 + 512 + 2048
    assert lib.pg_potrf_worksize(0, 16384) == 16384 * 128 + 16384 + 2048


def test_no_cpu_fallback_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    assert _ops._OPS is None
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pg.Exact_GP(torch.rand(8, 2, dtype=torch.float64), torch.rand(8, dtype=torch.float64), se_wn()).update()
    with pytest.raises(RuntimeError):
        pg.Squared_exponential().kernel(torch.ones(3, dtype=torch.float64), torch.rand(8, 2, dtype=torch.float64))
    src = open(os.path.join(os.path.dirname(_lib.__file__), "_ops.py")).read() + open(_lib.__file__).read()
    assert "oracle" not in src.replace("no CPU fallback", "")


def test_layout_of_composed_kernels():
    from pygpr_amd.covar import layout

    cov = pg.Compose([pg.White_noise(), pg.Squared_exponential(), pg.Compose([pg.Matern52(), pg.White_noise()])])
    kinds, offs, noise, nhp = layout(cov, 3)
    assert kinds == [_lib.PG_KIND_RBF, _lib.PG_KIND_MATERN52] and offs == [1, 5] and noise == [0, 9] and nhp == 10
    x = torch.rand(2, 7, 3, dtype=torch.float64)
    assert cov.get_params_shape(x) == [2, 10]
    hp = cov.init_params(x)
    assert hp.shape == (2, 10) and float(hp[0, 0]) == 1e-4 and float(hp[1, 9]) == 1e-4 and float(hp[0, 1]) == 1.0
    with pytest.raises(ValueError):
        _ops.make_spec([0] * 5, list(range(5)), [])
    passes = _ops.make_specs([0] * 6, list(range(0, 24, 4)), [24, 25, 26, 27, 28])      # 6 stationary + 5 noise: two passes
    assert [(p.ncomp, p.nnoise) for p in passes] == [(4, 4), (2, 1)] and passes[1].off[0] == 16 and passes[1].noise_off[0] == 28


def test_host_logic_against_golden(fake_ops, golden):
    g = golden("gp")
    gp = pg.Exact_GP(T(g["a_x"]), T(g["a_y"]), se_wn())
    gp.set_params(T(g["a_hp"]))
    mu, var = gp.predict(T(g["a_xp"]), var="diag")
    np.testing.assert_allclose(mu.numpy(), g["a_mu"], atol=1e-10)
    np.testing.assert_allclose(var.numpy(), g["a_var"], atol=1e-11)
    _, cov = gp.predict(T(g["a_xp"]), var="full")
    np.testing.assert_allclose(cov.numpy(), g["a_cov"], atol=1e-11)
    np.testing.assert_allclose(gp.krnchd.numpy(), g["a_chol"], atol=1e-11)
    np.testing.assert_allclose(gp.wt.numpy(), g["a_wt"], rtol=1e-8)
    loss, grad = pg.MLE(gp).loss_and_grad(g["a_hp"].copy())
    np.testing.assert_allclose(loss, g["a_loss2"], rtol=1e-10)
    np.testing.assert_allclose(grad, g["a_grad2"], rtol=1e-8, atol=1e-8)
    # memo: the same parameters are not evaluated twice; new parameters or new data are
    calls = []
    mle = pg.MLE(gp)
    orig = mle._evaluate_device
    mle._evaluate_device = lambda p, w, k=None, r=False: (calls.append((w, r)), orig(p, w, k, r))[1]
    g1 = mle.grad(g["a_hp"].copy()); l1 = mle.loss(g["a_hp"].copy()); g2 = mle.grad(g["a_hp"].copy())
    assert calls == [(True, False)] and np.array_equal(g1, g2) and float(l1) == float(mle.loss_and_grad(g["a_hp"].copy())[0])
    l3 = mle.loss(g["a_hp"] * 1.01)
    g3 = mle.grad(g["a_hp"] * 1.01)       # a loss-only result cannot serve a gradient request, but its factor can:
    assert calls == [(True, False), (False, False), (True, True)]      # the second call re-uses it (no build, no Cholesky)
    fresh = pg.MLE(gp)
    fresh.memoize = False
    l4, g4 = fresh.loss_and_grad(g["a_hp"] * 1.01)
    np.testing.assert_allclose(g3, g4, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(l3, l4, rtol=1e-12)
    # batched experts: shapes and squeeze rules
    gpc = pg.Exact_GP(T(g["c_x"]), T(g["c_y"]), se_wn())
    gpc.set_params(T(g["c_hp"]))
    mu, var = gpc.predict(T(g["c_xp"]), var="diag")
    np.testing.assert_allclose(mu.numpy(), g["c_mu"], atol=1e-10)
    np.testing.assert_allclose(var.numpy(), g["c_var"], atol=1e-11)
    lc, gc = pg.MLE(gpc).loss_and_grad(g["c_hp"].copy())
    np.testing.assert_allclose(lc, g["c_loss"], rtol=1e-10)
    np.testing.assert_allclose(gc, g["c_grad"], rtol=1e-8, atol=1e-8)
    # dirty flag, non-PD error mapping, data reassignment
    assert not gp.need_upd
    gp.set_params(T(np.array([np.nan, 1, 1, 1, 0.1])))
    assert gp.need_upd
    with pytest.raises(torch.linalg.LinAlgError, match="leading minor of order"):
        gp.update()
    gp.set_params(T(g["a_hp"]))
    gp.update()
    gp.y = T(g["a_y"] * 2.0)
    assert gp.need_upd
    np.testing.assert_allclose(gp.wt.numpy(), 2.0 * g["a_wt"], rtol=1e-8)


def test_grbcm_and_drivers_on_host(fake_ops, golden):
    g = golden("grbcm")
    p = "g0_"
    m = pg.GRBCM(T(g[p + "xl"]), T(g[p + "yl"]), T(g[p + "xg"]), T(g[p + "yg"]), se_wn())
    m.gpg.set_params(T(g[p + "hpg"]))
    m.gpl.set_params(T(g[p + "hpl"]))
    mu, var = m.predict(T(g[p + "xs"]), var="diag")
    np.testing.assert_allclose(mu.numpy(), g[p + "mu"], atol=1e-10)
    np.testing.assert_allclose(var.numpy(), g[p + "var"], atol=1e-11)
    np.testing.assert_allclose(m.beta.numpy(), g[p + "beta"], atol=1e-9)
    if int(g[p + "full_ok"]):
        m.gpl.set_params(T(np.broadcast_to(g[p + "hpg"], g[p + "hpl"].shape).copy()))
        mu_f, cov_f = m.predict(T(g[p + "xs"]), var="full")
        np.testing.assert_allclose(cov_f.numpy(), g[p + "cov_full"], rtol=1e-6, atol=1e-10)
        np.testing.assert_allclose(mu_f.numpy(), g[p + "mu_full"], rtol=1e-6, atol=1e-9)
    m.set_params(T(np.array([1.0, 1.0, 1.0, 1.0, 0.2])))      # shared-hp training through the stock CG driver (8f-2)
    gl = pg.GRBCM_MLE(m)
    f0 = float(gl.loss(m.params.numpy()))
    cgm = pg.CG(gl)
    cgm.args.update(maxiter=3, disp=False)
    cgm.minimize()
    assert float(cgm.res.fun) < f0 and torch.equal(m.gpl.params[0], m.gpg.params)
    gg = golden("gp")
    gpd = pg.Exact_GP(T(gg["d_x"]), T(gg["d_y"]), se_wn())
    gpd.set_params(T(gg["d_hp"]))
    cg = pg.CG(pg.MLE(gpd))
    cg.args.update(maxiter=5, disp=False)
    cg.minimize()
    np.testing.assert_allclose(cg.res.fun, gg["d_res_fun"], rtol=1e-6)
    gam = pg.get_learn_rate(T(gg["a_hp"]), pg.MLE(pg.Exact_GP(T(gg["a_x"]), T(gg["a_y"]), se_wn())), 1e-6)
    np.testing.assert_allclose(gam, gg["a_gamma"], rtol=1e-6)     # host logic over the oracle-backed double: the oracle's accuracy
    nm = pg.Nelder_Mead(pg.MLE(gpd))
    nm.args.update(maxiter=3, disp=False)
    before = gpd.params.clone()
    nm.minimize()                                   # does not converge in 3 iterations: no write-back (opt.py:111-114)
    assert torch.equal(gpd.params, before)


def test_quadratic_optimisers_mock_loss(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(5)
    dim = 5
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
    with pytest.raises(NotImplementedError):
        pg.Loss(None).loss(np.zeros(2))
    with pytest.raises(NotImplementedError):
        pg.GPR(None, None, None).update()


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


def test_samplers_and_partition_host(fake_ops, golden):
    _check_sampler(golden)


@pytest.mark.parametrize("case", __import__("surface_cases").ALL, ids=lambda f: f.__name__)
def test_surface_round2_host(fake_ops, golden, case):
    case(golden("surface2"))


def test_reference_export_list_is_importable():
    """Every name `PyGPR/__init__.py:1-7` exports exists here; the two the reference cannot run itself raise."""
    for name in ("GPR Exact_GP Squared_exponential Covar Compose White_noise Loss MLE Opt CG Nelder_Mead BFGS_Quad CG_Quad hessian "
                 "GRBCM log_likelihood_batched UNIFORM MATERN1 sample_gp cluster_samples euclidean_dist SK_WRAP").split():
        assert hasattr(pg, name), name
    with pytest.raises(NotImplementedError):
        pg.sample_gp()
    with pytest.raises(NotImplementedError):
        pg.log_likelihood_batched()
