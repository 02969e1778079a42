"""GPU parity tests of the individual C-ABI entry points (through ctypes) against NumPy / the CPU oracle.
fp64 tolerances are stated per test; fp32 runs are compared with the fp64 answer at 1e-3 class."""
import os
import numpy as np
import pytest
import torch

from oracle import pygpr_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from pygpr_amd._ops import get_ops

    return get_ops()


def dev(a, dtype=torch.float64):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype)


def host(t):
    return t.detach().cpu().double().numpy()


def spd(n, rng, cond_shift=1.0):
    a = rng.standard_normal((n, n))
    return a @ a.T / n + cond_shift * np.eye(n)


# --------------------------------------------------------------------------- GEMM core
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 2e-4)])
@pytest.mark.parametrize("variant", ["NT", "NN", "TN", "TT"])
def test_gemm_variants(ops, variant, dtype, tol):
    from pygpr_amd._lib import GEMM_NN, GEMM_NT, GEMM_TN, GEMM_TT

    rng = np.random.default_rng(0)
    m, n, k = 256, 384, 208
    opa, opb = rng.standard_normal((m, k)), rng.standard_normal((k, n))
    c0 = rng.standard_normal((m, n))
    code = {"NT": GEMM_NT, "NN": GEMM_NN, "TN": GEMM_TN, "TT": GEMM_TT}[variant]
    a = dev(opa.T if variant[0] == "T" else opa, dtype)
    b = dev(opb.T if variant[1] == "T" else opb, dtype)
    c = dev(c0, dtype)
    ops.gemm_raw(code, m, n, k, -0.5, a, b, 2.0, c)
    ref = -0.5 * opa @ opb + 2.0 * c0
    np.testing.assert_allclose(host(c), ref, atol=tol * k, rtol=0)
    # asymmetric-B identity check (guards a swapped C/D register map)
    eye = dev(np.eye(256), dtype)
    basym = rng.standard_normal((256, 256))
    c = dev(np.zeros((256, 256)), dtype)
    ops.gemm_raw(GEMM_NN, 256, 256, 256, 1.0, eye, dev(basym, dtype), 0.0, c)
    np.testing.assert_allclose(host(c), basym, atol=1e-6 if dtype == torch.float32 else 0)


def test_gemm_row_panel_inplace_and_tri(ops):
    from pygpr_amd._lib import GEMM_NT, GEMM_NT_RP

    rng = np.random.default_rng(1)
    # in-place B <- B inv^T with a lower-triangular inv (khi = 2 skips k > column)
    m = 320
    bmat = rng.standard_normal((m, 256))
    inv = np.tril(rng.standard_normal((256, 256)))
    b = dev(bmat)
    ops.gemm_raw(GEMM_NT_RP, m - m % 64, 256, 256, 1.0, b, dev(inv), 0.0, b, khi=2)
    np.testing.assert_allclose(host(b), bmat @ inv.T, atol=1e-11)
    # SYRK on lower tiles only: tiles above the diagonal stay untouched
    n, k = 512, 256
    p = rng.standard_normal((n, k))
    c0 = rng.standard_normal((n, n))
    c = dev(c0)
    pd = dev(p)
    ops.gemm_raw(GEMM_NT, n, n, k, -1.0, pd, pd, 1.0, c, tri=1)
    got, ref = host(c), c0 - p @ p.T
    for ti in range(n // 128):
        for tj in range(n // 128):
            blk = (slice(ti * 128, ti * 128 + 128), slice(tj * 128, tj * 128 + 128))
            np.testing.assert_allclose(got[blk], ref[blk] if tj <= ti else c0[blk], atol=1e-11)


def test_gemm_skinny_chain_variants(ops):
    """The Cholesky chain's two products in their 64-row and 32-row tilings: C -= A B^T with a 128-wide C, and the
    in-place panel solve B <- B inv^T against a lower-triangular 128 x 128 inverse (khi = 2)."""
    from pygpr_amd._lib import GEMM_NT_32x64, GEMM_NT_32x128, GEMM_NT_64, GEMM_NT_64x128

    rng = np.random.default_rng(21)
    m, k = 416, 384                      # 13 row tiles of 32: ragged against 64
    a = rng.standard_normal((m, k))
    b = rng.standard_normal((128, k))
    c0 = rng.standard_normal((m, 128))
    inv = np.tril(rng.standard_normal((128, 128)))
    for vu, vt, mm in ((GEMM_NT_64, GEMM_NT_64x128, 384), (GEMM_NT_32x64, GEMM_NT_32x128, 416)):
        c = dev(c0[:mm])
        ops.gemm_raw(vu, mm, 128, k, -1.0, dev(a[:mm]), dev(b), 1.0, c)
        np.testing.assert_allclose(host(c), c0[:mm] - a[:mm] @ b.T, atol=1e-11)
        t = dev(c0[:mm])
        ops.gemm_raw(vt, mm, 128, 128, 1.0, t, dev(inv), 0.0, t, khi=2)
        np.testing.assert_allclose(host(t), c0[:mm] @ inv.T, atol=1e-11)
    cf = dev(c0[:416], torch.float32)
    ops.gemm_raw(GEMM_NT_32x64, 416, 128, k, -1.0, dev(a, torch.float32), dev(b, torch.float32), 1.0, cf)
    np.testing.assert_allclose(host(cf), c0 - a @ b.T, atol=2e-3)
    # the tail's U product: 32 x 32 tiles, 64-deep K tile
    from pygpr_amd._lib import GEMM_NT_32x32
    for dt, tol in ((torch.float64, 1e-11), (torch.float32, 2e-3)):
        c = dev(c0, dt)
        ops.gemm_raw(GEMM_NT_32x32, 416, 128, k, -1.0, dev(a, dt), dev(b, dt), 1.0, c)
        np.testing.assert_allclose(host(c), c0 - a @ b.T, atol=tol)


def test_syrk_tn_quarter_tiles_same_bits(ops):
    """K** - V^T V on 64 x 64 tiles (GEMM_TN_64: outputs whose 128 x 128 tiles do not fill the chip) gives the bits of the 128 x 128
    launch: an element's k order does not depend on the tile shape; pg_syrk_tn_sub picks the quarter tiles by itself."""
    from pygpr_amd._lib import GEMM_TN, GEMM_TN_64

    rng = np.random.default_rng(17)
    m, k = 384, 272
    v = rng.standard_normal((k, m))
    c0 = rng.standard_normal((m, m))
    for dt, tol in ((torch.float64, 1e-11), (torch.float32, 2e-3)):
        outs = []
        for variant in (GEMM_TN, GEMM_TN_64):
            c = dev(c0, dt)
            ops.gemm_raw(variant, m, m, k, -1.0, dev(v, dt), dev(v, dt), 1.0, c, tri=1)
            outs.append(host(c))
        low = np.tril_indices(m)
        np.testing.assert_allclose(outs[1][low], (c0 - v.T @ v)[low], atol=tol)
        assert np.array_equal(outs[0][low], outs[1][low])
        assert np.array_equal(np.triu(outs[1], 128), np.triu(host(dev(c0, dt)), 128))   # tiles above the diagonal untouched
        c = dev(c0, dt)
        ops.syrk_tn_sub(dev(v, dt), c, lower_only=True)
        assert np.array_equal(host(c)[low], outs[0][low])


def test_gemm_triangular_k_ranges(ops):
    from pygpr_amd._lib import GEMM_NN, GEMM_TN

    rng = np.random.default_rng(2)
    n = 512
    low = np.tril(rng.standard_normal((n, n)))
    dense = rng.standard_normal((n, n))
    # khi = 1: lower-triangular left operand
    c = dev(np.zeros((n, n)))
    ops.gemm_raw(GEMM_NN, n, n, n, 1.0, dev(low), dev(dense), 0.0, c, khi=1)
    np.testing.assert_allclose(host(c), low @ dense, atol=1e-11)
    # klo = 2: lower-triangular right operand
    c = dev(np.zeros((n, n)))
    ops.gemm_raw(GEMM_NN, n, n, n, 1.0, dev(dense), dev(low), 0.0, c, klo=2)
    np.testing.assert_allclose(host(c), dense @ low, atol=1e-11)
    # klo = 1 on lower tiles: low^T low
    c = dev(np.zeros((n, n)))
    ld = dev(low)
    ops.gemm_raw(GEMM_TN, n, n, n, 1.0, ld, ld, 0.0, c, tri=1, klo=1)
    np.testing.assert_allclose(np.tril(host(c)), np.tril(low.T @ low), atol=1e-11)


# --------------------------------------------------------------------------- covariance build
def _spec(covs, d):
    from pygpr_amd._lib import PG_KIND_MATERN52, PG_KIND_RBF
    from pygpr_amd._ops import make_spec

    kinds, offs, noise, o = [], [], [], 0
    for c in covs:
        if c.kind == "wn":
            noise.append(o)
            o += 1
        else:
            kinds.append(PG_KIND_RBF if c.kind == "se" else PG_KIND_MATERN52)
            offs.append(o)
            o += d + 1
    return make_spec(kinds, offs, noise)


@pytest.mark.parametrize("covs", [[orc.SE, orc.WN], [orc.SE, orc.SE, orc.WN], [orc.M52, orc.WN], [orc.WN, orc.SE]])
@pytest.mark.parametrize("n,d", [(10, 2), (200, 5), (300, 16)])
def test_kernel_build(ops, covs, n, d):
    from pygpr_amd._ops import pad_to

    rng = np.random.default_rng(n + d)
    x, xp = rng.random((n, d)), rng.random((37, d))
    hp = np.concatenate([0.5 + rng.random(c.nhp(d)) if c.kind != "wn" else 0.05 + 0.1 * rng.random(1) for c in covs])
    spec, npad, mpad = _spec(covs, d), pad_to(n), pad_to(37)
    hpd, xd, xpd = dev(hp), dev(x), dev(xp)
    k = ops.empty(npad, npad)
    ops.kernel_build(spec, hpd, xd, None, k, jitter=1e-7)
    ref = orc.kernel(covs, hp, x, form="direct") + 1e-7 * np.eye(n)
    got = host(k)
    np.testing.assert_allclose(got[:n, :n], ref, atol=1e-14, rtol=1e-14)
    assert np.array_equal(got[:n, :n], got[:n, :n].T)            # exactly symmetric
    pad_ref = np.eye(npad)
    pad_ref[:n, :n] = got[:n, :n]
    assert np.array_equal(got, pad_ref)                           # identity padding
    # cross build, train rows x test columns (transposed w.r.t. the reference's [m, n])
    ks = ops.empty(npad, mpad)
    ops.kernel_build(spec, hpd, xd, xpd, ks)
    ref_ks = orc.kernel(covs, hp, x, xp, form="direct").T
    got = host(ks)
    np.testing.assert_allclose(got[:n, :37], ref_ks, atol=1e-14, rtol=1e-14)
    assert not got[n:, :].any() and not got[:, 37:].any()
    # lower-only build leaves tiles above the diagonal untouched
    kl = ops.zeros(npad, npad) - 7.0
    ops.kernel_build(spec, hpd, xd, None, kl, lower_only=True, jitter=1e-7)
    gl = host(kl)
    np.testing.assert_array_equal(np.tril(gl), np.tril(host(k)))
    if npad >= 128:
        assert (gl[:64, 64:128] == -7.0).all()


@pytest.mark.parametrize("d", [31, 48, 64])
def test_kernel_build_large_d(ops, d):
    """From d = 31 the mirrored fp64 build needs more than the 64 KB of LDS a kernel gets without opting in (101 KB at
    d = 64 = PG_MAX_DIM): full, lower-only and cross builds, and the NLML gradient at the same d, against the oracle."""
    from pygpr_amd._ops import pad_to

    rng = np.random.default_rng(d)
    n, m = 200, 70
    covs = [orc.SE, orc.WN]
    x, xp = rng.random((n, d)), rng.random((m, d))
    hp = np.concatenate([[1.1], (0.2 + 0.3 * rng.random(d)) / np.sqrt(d), [0.1]])
    spec, npad, mpad = _spec(covs, d), pad_to(n), pad_to(m)
    hpd, xd, xpd = dev(hp), dev(x), dev(xp)
    ref = orc.kernel(covs, hp, x, form="direct") + 1e-7 * np.eye(n)
    k = ops.empty(npad, npad)
    ops.kernel_build(spec, hpd, xd, None, k, jitter=1e-7)
    got = host(k)
    np.testing.assert_allclose(got[:n, :n], ref, atol=1e-14, rtol=1e-14)
    assert np.array_equal(got, got.T) and np.array_equal(got[n:, n:], np.eye(npad - n))
    kl = ops.zeros(npad, npad) - 7.0
    ops.kernel_build(spec, hpd, xd, None, kl, lower_only=True, jitter=1e-7)
    np.testing.assert_array_equal(np.tril(host(kl)), np.tril(got))
    assert (host(kl)[:64, 64:128] == -7.0).all()
    ks = ops.empty(npad, mpad)
    ops.kernel_build(spec, hpd, xd, xpd, ks)
    np.testing.assert_allclose(host(ks)[:n, :m], orc.kernel(covs, hp, x, xp, form="direct").T, atol=1e-14, rtol=1e-14)
    kf = ops.empty(npad, npad, dtype=torch.float32)
    ops.kernel_build(spec, hpd, dev(x, torch.float32), None, kf, jitter=1e-7)
    np.testing.assert_allclose(host(kf)[:n, :n], ref, atol=5e-6)


def test_kernel_build_passes_accumulate(ops):
    """A Compose with more children than one pg_covspec holds is evaluated in passes (make_specs): the first pass writes,
    later ones accumulate and leave the padding alone; gradient passes fill disjoint entries."""
    from pygpr_amd._ops import make_specs, pad_to
    from pygpr_amd._lib import PG_KIND_MATERN52, PG_KIND_RBF

    rng = np.random.default_rng(3)
    n, d = 150, 3
    covs = [orc.SE, orc.WN, orc.M52, orc.SE, orc.WN, orc.WN, orc.SE, orc.WN, orc.SE, orc.M52, orc.WN]
    kinds, offs, noise, o = [], [], [], 0
    for c in covs:
        if c.kind == "wn":
            noise.append(o); o += 1
        else:
            kinds.append(PG_KIND_RBF if c.kind == "se" else PG_KIND_MATERN52); offs.append(o); o += d + 1
    specs = make_specs(kinds, offs, noise)
    assert len(specs) == 2
    hp = np.concatenate([0.4 + 0.4 * rng.random(c.nhp(d)) if c.kind != "wn" else 0.05 + 0.1 * rng.random(1) for c in covs])
    x, y = rng.random((n, d)), rng.standard_normal(n)
    npad = pad_to(n)
    k = ops.empty(npad, npad)
    ops.kernel_build(specs, dev(hp), dev(x), None, k, jitter=1e-7)
    got = host(k)
    np.testing.assert_allclose(got[:n, :n], orc.kernel(covs, hp, x, form="direct") + 1e-7 * np.eye(n), atol=1e-13)
    assert np.array_equal(got[n:, n:], np.eye(npad - n)) and not got[n:, :n].any()
    ks = ops.empty(npad, 256)
    xp = rng.random((40, d))
    ops.kernel_build(specs, dev(hp), dev(x), dev(xp), ks)
    np.testing.assert_allclose(host(ks)[:n, :40], orc.kernel(covs, hp, x, xp, form="direct").T, atol=1e-13)
    # gradient through the passes, against the oracle's K^-1 route
    kd = orc.kernel(covs, hp, x, form="direct") + 1e-7 * np.eye(n)
    kinv = np.linalg.inv(kd)
    alpha = kinv @ y
    kpad = np.eye(npad); kpad[:n, :n] = kinv
    apad = np.zeros(npad); apad[:n] = alpha
    grad = ops.zeros(hp.size)
    work = ops.empty(ops.nlml_grad_worksize(n, hp.size))
    ops.nlml_grad(specs, dev(hp), dev(x), n, dev(kpad), dev(apad), grad, work)
    _, gref = orc.mle_loss_and_grad(covs, hp, x, y, "kinv", form="direct")
    np.testing.assert_allclose(host(grad), gref, rtol=1e-8, atol=1e-8 * np.abs(gref).max())


def test_sqdist_kind_and_centres_any_d(ops):
    """PG_KIND_SQDIST through the covariance tile kernel (Squared_exponential.distance), and pg_sqdist_argmin with more
    centres x dimensions than 64 KB of LDS holds at once (m = 1500, d = 16 in fp64; d = 64)."""
    rng = np.random.default_rng(8)
    x, c = rng.random((300, 5)), rng.random((90, 5))
    out = ops.empty(320, 128)
    ops.sqdist(dev(x), dev(c), out)
    np.testing.assert_allclose(host(out)[:300, :90], ((x[:, None, :] - c[None, :, :]) ** 2).sum(2), atol=1e-14)
    for m, d in ((1500, 16), (200, 64)):
        x, c = rng.random((700, d)), rng.random((m, d))
        d2 = ((x[:, None, :] - c[None, :, :]) ** 2).sum(2)
        dist = ops.empty(700, m)
        idx = torch.zeros(700, dtype=torch.int32, device="cuda")
        ops.sqdist_argmin(dev(x), dev(c), dist, idx)
        np.testing.assert_allclose(host(dist), d2, atol=1e-12)
        assert np.array_equal(idx.cpu().numpy(), np.argmin(d2, axis=1))


def test_kernel_build_propagates_nan(ops):
    """A NaN coordinate must reach K as NaN (torch's exp does; the reference's Cholesky then raises): the device exp must
    not launder it into a finite value."""
    rng = np.random.default_rng(4)
    n, d = 70, 3
    x = rng.random((n, d))
    x[23, 1] = np.nan
    hp = np.array([1.0, 0.7, 0.8, 0.9, 0.1])
    for dt in (torch.float64, torch.float32):
        k = ops.empty(256, 256, dtype=dt)
        ops.kernel_build(_spec([orc.SE, orc.WN], d), dev(hp), dev(x, dt), None, k, jitter=1e-7)
        got = host(k)
        assert np.isnan(got[23, :n]).all() and np.isnan(got[:n, 23]).all()
        assert np.isfinite(np.delete(np.delete(got[:n, :n], 23, 0), 23, 1)).all()
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        ops.potrf(k, ops.potrf_workspace(256, dt), info)
        assert int(info.item()) == 24            # LAPACK convention: the leading minor of order 24


@pytest.mark.parametrize("offset", [0.0, 1.0e3, 1.0e4])
def test_kernel_build_fast_body_on_uncentred_data(ops, offset):
    """The fp64 one-squared-exponential build forms the exponent as 2 x.x' - |x|^2 - |x'|^2 (the reference's own expansion,
    covar.py:102-127): its ABSOLUTE error is eps |x l|^2, whatever the distance.  With the points shifted by 1e3 / 1e4 (|x l|^2 ~ 3e6 /
    3e8) the covariance must stay within that bound of the direct-difference oracle, never exceed sigma^2 (the argument is clamped at 0:
    near-duplicate points), and the diagonal of a symmetric build is sigma^2 + sigma_n^2 + jitter EXACTLY; the matrix still factorises."""
    rng = np.random.default_rng(17)
    n, d = 700, 3
    x = rng.random((n, d))
    x[1] = x[0] + 1e-9                      # a near-duplicate pair
    x += offset
    hp = np.array([1.3, 1.0, 0.9, 1.1, 0.05])
    covs = [orc.SE, orc.WN]
    k = ops.empty(768, 768)
    ops.kernel_build(_spec(covs, d), dev(hp), dev(x), None, k, jitter=1e-7)
    got = host(k)[:n, :n]
    ref = orc.kernel(covs, hp, x, form="direct") + 1e-7 * np.eye(n)
    bound = 2.0 ** -52 * 8.0 * (d * (offset + 1.0) ** 2 * 1.21) * 1.69 + 1e-14      # eps x (a few) x |x l|^2 x sigma^2
    np.testing.assert_allclose(got, ref, atol=bound, rtol=0)
    diag = 1.3 ** 2 + 0.05 ** 2 + 1e-7
    assert np.all(np.diag(got) == diag) and got[np.triu_indices(n, 1)].max() <= 1.3 ** 2
    np.testing.assert_array_equal(got, got.T)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.potrf(k, ops.potrf_workspace(768, torch.float64), info)
    assert int(info.item()) == 0


def test_kernel_build_fp32(ops):
    from pygpr_amd._ops import pad_to

    rng = np.random.default_rng(9)
    n, d = 300, 8
    x = rng.random((n, d))
    covs = [orc.M52, orc.WN]
    hp = np.concatenate([[1.2], 0.5 + rng.random(d), [0.1]])
    k = ops.empty(pad_to(n), pad_to(n), dtype=torch.float32)
    ops.kernel_build(_spec(covs, d), dev(hp), dev(x, torch.float32), None, k, jitter=1e-7)
    np.testing.assert_allclose(host(k)[:n, :n], orc.kernel(covs, hp, x) + 1e-7 * np.eye(n), atol=5e-6)


# --------------------------------------------------------------------------- Cholesky & friends
@pytest.mark.parametrize("n", [256, 768, 1280])
def test_potrf_solves_inverse(ops, n):
    rng = np.random.default_rng(n)
    a = spd(n, rng)
    ad = dev(a)
    invd = ops.potrf_workspace(n, torch.float64)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.potrf(ad, invd, info)
    assert int(info.item()) == 0
    chol = np.linalg.cholesky(a)
    np.testing.assert_allclose(np.tril(host(ad)), chol, atol=1e-12)
    # alpha = A^-1 y
    y = rng.standard_normal(n)
    xd = ops.empty(n)
    ops.potrs_vec(ad, invd, dev(y), xd)
    np.testing.assert_allclose(host(xd), np.linalg.solve(a, y), rtol=1e-9, atol=1e-10)
    # Minv = L^-1
    minv = ops.zeros(n, n)
    ops.trtri(ad, invd, minv)
    linv = np.linalg.inv(chol)
    got = host(minv)
    for ti in range(n // 128):   # strictly upper 128-tiles are scratch of the recursive doubling
        for tj in range(ti + 1):
            blk = (slice(ti * 128, ti * 128 + 128), slice(tj * 128, tj * 128 + 128))
            np.testing.assert_allclose(got[blk], linv[blk], atol=1e-11)
    # trmv both ways
    v = rng.standard_normal(n)
    out = ops.empty(n)
    work = ops.empty((n // 256) * n)
    ops.trmv(minv, dev(v), out, 0)
    np.testing.assert_allclose(host(out), linv @ v, atol=1e-11)
    ops.trmv(minv, dev(v), out, 1, work)
    np.testing.assert_allclose(host(out), linv.T @ v, atol=1e-11)
    # Kinv = Minv^T Minv
    kinv = ops.zeros(n, n)
    ops.lauum(minv, kinv)
    np.testing.assert_allclose(np.tril(host(kinv)), np.tril(np.linalg.inv(a)), atol=1e-10)
    # tril export
    ops.tril(ad, n)
    np.testing.assert_allclose(host(ad), chol, atol=1e-12)


@pytest.mark.parametrize("n", [3072, 2816])
def test_potrf_lookahead_matches_sequential(ops, n):
    """n = 3072 spans six 512-column outer panels (2816: the last one ragged), so the two-stream look-ahead schedule
    is active; it must give the same factor as the single-stream schedule (to rounding: see below) and
    match LAPACK."""
    rng = np.random.default_rng(33)
    a = spd(n, rng)
    outs = []
    for la in (1, 0):
        ops.set_lookahead(la)
        ad = dev(a)
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        ops.potrf(ad, ops.potrf_workspace(n, torch.float64), info)
        assert int(info.item()) == 0
        outs.append(np.tril(host(ad)))
    ops.set_lookahead(1)
    # the flag-coupled chain of the look-ahead schedule (chainstep.hip) sums a block column's update over two panels at once:
    # the same terms in another order, so the two factors agree to rounding, not bit for bit
    np.testing.assert_allclose(outs[0], outs[1], rtol=0, atol=2e-13)
    np.testing.assert_allclose(outs[0], np.linalg.cholesky(a), atol=1e-11)


@pytest.mark.parametrize("panel,n", [(1024, 3072), (1024, 3328), (2048, 6144), (256, 1280)])
def test_potrf_outer_panel_widths(ops, panel, n):
    """The outer panel is chosen from n (512 / 1024 / 2048); every width, forced through pg_set_outer_panel, must give the
    LAPACK factor, with and without look-ahead."""
    rng = np.random.default_rng(panel + n)
    a = spd(n, rng)
    chol = np.linalg.cholesky(a)
    ops.set_outer_panel(panel)
    try:
        outs = []
        for la in (1, 0):
            ops.set_lookahead(la)
            ad = dev(a)
            info = torch.zeros(1, dtype=torch.int32, device="cuda")
            ops.potrf(ad, ops.potrf_workspace(n, torch.float64), info)
            assert int(info.item()) == 0
            outs.append(np.tril(host(ad)))
        np.testing.assert_allclose(outs[0], outs[1], rtol=0, atol=2e-13)   # to rounding: the coupled chain sums in another order
        np.testing.assert_allclose(outs[0], chol, atol=1e-11)
        # fused inverse with this width
        ad = dev(a)
        ops.set_lookahead(1)
        minv = ops.zeros(n, n)
        ops.potrf_trtri(ad, ops.potrf_workspace(n, torch.float64), info, minv)
        v = rng.standard_normal(n)
        np.testing.assert_allclose(np.tril(host(minv)) @ (chol @ v), v, atol=1e-9)
    finally:
        ops.set_outer_panel(0)
        ops.set_lookahead(1)


@pytest.mark.parametrize("n", [8192, 5376])
def test_potrf_trtri_fused_matches_separate(ops, n):
    """pg_potrf_trtri = pg_potrf followed by pg_trtri on the caller's stream (the background-stream overlap of round 1 is off by
    default, PG_BG_STREAM=1; 8192: every panel on the coupled chain; 5376: the last panel ragged, trailing part not a power
    of two).  The factor must equal the separate call bit for bit, the inverse too when both routes pair the diagonal
    blocks the same way; and the inverse must undo the factor."""
    g = torch.Generator(device="cuda").manual_seed(44)
    r = torch.randn(n, n, device="cuda", dtype=torch.float64, generator=g)
    a = torch.mm(r, r.T) / n + torch.eye(n, device="cuda", dtype=torch.float64)      # test data only
    del r
    ad, bd = a.clone(), a.clone()
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    invd = ops.potrf_workspace(n, torch.float64)
    m1, m2 = ops.zeros(n, n), ops.zeros(n, n)
    ops.potrf_trtri(ad, invd, info, m1)
    assert int(info.item()) == 0
    invd2 = ops.potrf_workspace(n, torch.float64)
    ops.potrf(bd, invd2, info)
    ops.trtri(bd, invd2, m2)
    l1, l2 = torch.tril(ad), torch.tril(bd)
    assert torch.equal(l1, l2)
    g1, g2 = torch.tril(m1), torch.tril(m2)
    if n & (n - 1) == 0:     # power of two: both routes pair the diagonal blocks identically
        assert torch.equal(g1, g2)
    else:                    # the split pairs them differently: same inverse, different rounding order
        assert float((g1 - g2).abs().max()) <= 1e-12 * float(g2.abs().max())
    v = torch.randn(n, device="cuda", dtype=torch.float64, generator=g)
    assert float((g1 @ (l1 @ v) - v).abs().max()) <= 1e-9                          # L^-1 (L v) = v
    assert float((l1 @ (l1.T @ v) - a @ v).abs().max()) <= 1e-9 * float((a @ v).abs().max())   # L L^T = A


@pytest.mark.parametrize("n,with_build,dtype", [(1024, False, torch.float64), (1536, True, torch.float64), (2560, False, torch.float64),
                                                 (2048, True, torch.float32)])
def test_recursive_split_matches_one_level_schedule(ops, n, with_build, dtype):
    """Round 4: from pg_set_recursive_split's size on (default 16384) the fused factor-and-invert call splits the matrix at n / 2 and
    computes the blocks that cross the split as products against the leading half's inverse (linalg.hip: potrf_trtri_rec).  Forced
    at small sizes (1024: one level; 1536: halves of 768; 2560: TWO levels, 1280 -> 768 + 512; fp32), with a caller-supplied matrix
    (the off-diagonal block is copied into the inverse's buffer) and with the folded covariance build (it is built there): factor and
    inverse against the one-level schedule of the same library and against LAPACK."""
    from scipy.linalg import lapack
    rng = np.random.default_rng(n)
    tol = 1e-11 if dtype == torch.float64 else 2e-3
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    res = []
    try:
        for rec in (512, 0):
            ops.set_recursive_split(rec)
            a, m = ops.empty(n, n, dtype=dtype), ops.zeros(n, n, dtype=dtype)
            invd = ops.potrf_workspace(n, dtype)
            if with_build:
                d = 3
                x = np.random.default_rng(7).random((n - 37, d))         # 37 rows of identity padding in the trailing block
                hp = np.array([1.1, 0.8, 0.9, 1.2, 0.3])
                ops.build_factor(_spec([orc.SE, orc.WN], d), dev(hp), dev(x, dtype), a, invd, info, m)
                k = orc.kernel([orc.SE, orc.WN], hp, x) + 1e-7 * np.eye(n - 37)
                ref = np.eye(n)
                ref[:n - 37, :n - 37] = k
            else:
                ref = spd(n, rng) if rec else ref
                a.copy_(dev(ref, dtype))
                ops.potrf_trtri(a, invd, info, m)
            assert int(info.item()) == 0
            res.append((np.tril(host(a)), np.tril(host(m))))
        (l_rec, m_rec), (l_one, m_one) = res
        want, _ = lapack.dpotrf(ref, lower=1)
        np.testing.assert_allclose(l_rec, np.tril(want), atol=tol)
        np.testing.assert_allclose(l_rec, l_one, atol=tol)
        np.testing.assert_allclose(m_rec, m_one, atol=tol * 10 * np.abs(m_one).max())
        np.testing.assert_allclose(m_rec @ l_rec, np.eye(n), atol=tol * 100)
    finally:
        ops.set_recursive_split(16384)


@pytest.mark.parametrize("j", [5, 300, 511, 512, 700, 1023])
def test_recursive_split_names_the_first_bad_minor(ops, j):
    """A non-positive pivot in the leading half (later launches return at once, the trailing half must not reset the flag), on the
    split, and in the trailing half (the leaf's column + the block's offset): LAPACK's info = j + 1 either way."""
    from scipy.linalg import lapack
    rng = np.random.default_rng(90 + j)
    n = 1024
    a = spd(n, rng)
    a[j, j] = -0.5
    _, want = lapack.dpotrf(a, lower=1)
    assert want == j + 1
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    try:
        ops.set_recursive_split(512)
        ops.potrf_trtri(dev(a), ops.potrf_workspace(n, torch.float64), info, ops.empty(n, n))
        assert int(info.item()) == want
    finally:
        ops.set_recursive_split(16384)


@pytest.mark.parametrize("t,tri", [(33, 1), (24, 0), (46, 1)])
def test_gemm_mixed_launch_is_bit_identical(ops, t, tri):
    """Round 4: a launch of equally long 128 x 128 tiles that does not fill a whole number of rounds of the chip's workgroup slots ends in
    64 x 64 quarter tiles (pg_gemm_mixed_kernel; 561 / 576 / 1081 tiles here: 49, 64 and 57 past a round of 512).  Every output element is
    still one workgroup's sum over k in the same order: against the same product through 128-tile launches that are NOT mixed
    (row slabs of at most 512 tiles), bit for bit; and against NumPy."""
    from pygpr_amd._lib import GEMM_NT

    g = torch.Generator(device="cuda").manual_seed(t)
    n, k = 128 * t, 256
    a = torch.randn(n, k, device="cuda", dtype=torch.float64, generator=g)
    c0 = torch.randn(n, n, device="cuda", dtype=torch.float64, generator=g)
    c1, c2 = c0.clone(), c0.clone()
    ops.gemm_raw(GEMM_NT, n, n, k, -1.0, a, a, 1.0, c1, tri=tri)
    rows = 128 * (512 // t)                                   # slabs of fewer than 512 tiles: never mixed
    for r0 in range(0, n, rows):
        r1 = min(n, r0 + rows)
        ops.gemm_raw(GEMM_NT, r1 - r0, n, k, -1.0, a[r0:r1], a, 1.0, c2[r0:r1])
    m1, m2 = (torch.tril(c1), torch.tril(c2)) if tri else (c1, c2)
    assert torch.equal(m1, m2)
    ref = host(c0) - host(a) @ host(a).T
    np.testing.assert_allclose(host(m1), np.tril(ref) if tri else ref, atol=1e-11)
    if tri:     # tiles strictly above the diagonal are not touched
        i, jj = 0, n - 128
        assert torch.equal(c1[i:i + 128, jj:jj + 128], c0[i:i + 128, jj:jj + 128])


@pytest.mark.parametrize("n", [1792, 2304, 3072, 4864])
def test_potrs_vec_blocked_sweeps(ops, n):
    """x = A^-1 y from the factor.  Below n = 2048 one fused step per 128 columns; above, sweeps over 1024-wide blocks
    against their inverses (2304 = two blocks + a 256 remainder, 4864 = four + 768)."""
    rng = np.random.default_rng(n)
    a = spd(n, rng)
    y = rng.standard_normal(n)
    ad = dev(a)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    invd = ops.potrf_workspace(n, torch.float64)
    ops.potrf(ad, invd, info)
    assert int(info.item()) == 0
    x = ops.empty(n)
    yd = dev(y)
    ops.potrs_vec(ad, invd, yd, x)
    assert torch.equal(yd, dev(y))                        # the right-hand side is not modified
    ref = np.linalg.solve(a, y)
    np.testing.assert_allclose(host(x), ref, rtol=0, atol=1e-11 * np.abs(ref).max())
    a32 = dev(a, torch.float32)
    invd32 = ops.potrf_workspace(n, torch.float32)
    ops.potrf(a32, invd32, info)
    x32 = ops.empty(n, dtype=torch.float32)
    ops.potrs_vec(a32, invd32, dev(y, torch.float32), x32)
    np.testing.assert_allclose(host(x32), ref, rtol=0, atol=5e-4 * np.abs(ref).max())


def test_bad_arguments_are_refused_with_a_message(ops):
    """Error convention of the C ABI: a negative status and a text from pg_last_error, nothing launched."""
    from pygpr_amd import _lib

    n = 300                                   # not a multiple of 256
    a = ops.zeros(n, n)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match="multiple of 256"):
        ops.potrf(a, ops.potrf_workspace(512, torch.float64), info)
    with pytest.raises(RuntimeError, match="not tile aligned"):
        ops.gemm_raw(_lib.GEMM_NT, 200, 128, 64, 1.0, ops.zeros(200, 64), ops.zeros(128, 64), 0.0, ops.zeros(200, 128))
    with pytest.raises(RuntimeError, match="outer panel"):
        ops.set_outer_panel(100)
    assert _lib.load().pg_potrf(None, 0, 256, None, 256, None, None, None) < 0      # null handle / pointers
    assert b"null" in _lib.load().pg_last_error()


def test_potri_and_logdet(ops):
    """pg_potri = K^-1 from the factor in one call (may overwrite the factor); pg_logdet = 2 sum log L_ii over the real rows."""
    rng = np.random.default_rng(71)
    n = 768
    a = spd(n, rng)
    ad = dev(a)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    invd = ops.potrf_workspace(n, torch.float64)
    ops.potrf(ad, invd, info)
    out = ops.zeros(1)
    ops.logdet(ad, n, out)
    np.testing.assert_allclose(float(out[0]), np.linalg.slogdet(a)[1], rtol=1e-12)
    ops.logdet(ad, 500, out)                                   # leading 500 rows only (padding excluded by the caller)
    np.testing.assert_allclose(float(out[0]), 2.0 * np.log(np.diag(np.linalg.cholesky(a))[:500]).sum(), rtol=1e-12)
    ops.potri(ad, invd, ad)                                    # in place over the factor
    ref = np.linalg.inv(a)
    np.testing.assert_allclose(np.tril(host(ad)), np.tril(ref), rtol=0, atol=1e-11 * np.abs(ref).max())


def test_potrf_not_positive_definite_reports_minor(ops):
    rng = np.random.default_rng(4)
    n = 512
    a = spd(n, rng)
    a[300, 300] = -1.0   # leading minor of order 301 fails
    ad = dev(a)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.potrf(ad, ops.potrf_workspace(n, torch.float64), info)
    assert int(info.item()) == 301


@pytest.mark.parametrize("j", [0, 1, 3, 4, 7, 8, 12, 15, 16, 21, 63, 127, 128, 133, 255, 300, 511])
def test_potrf_names_the_first_bad_minor_at_every_position(ops, j):
    """The leaf factors its 16 x 16 diagonal blocks four columns at a time with the rank-4 updates on the matrix pipe (leaf3_factor_blk):
    a non-positive pivot in any lane group, panel or block must come back as LAPACK's info = j + 1 (torch.cholesky's message names that
    minor, gpr.py:69), and a second bad pivot further down must not mask it."""
    from scipy.linalg import lapack
    rng = np.random.default_rng(40 + j)
    n = 512
    a = spd(n, rng)
    a[j, j] = -0.5
    if j + 9 < n:
        a[j + 9, j + 9] = -2.0
    _, want = lapack.dpotrf(a, lower=1)
    assert want == j + 1
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.potrf(dev(a), ops.potrf_workspace(n, torch.float64), info)
    assert int(info.item()) == want


def test_leaf_on_badly_scaled_tiles_matches_lapack(ops):
    """One 128 x 128 leaf (factor + inverse) on a tile whose rows span twelve orders of magnitude and on a tile of condition 1e10: the
    blocked factor's square-root-free elimination, its staging through LDS and the scaling at the end, against LAPACK and against the
    backward-error bounds a Cholesky factor and a triangular inverse must meet whatever the conditioning."""
    rng = np.random.default_rng(77)
    for kind in ("scaled", "illcond"):
        if kind == "scaled":
            q = rng.standard_normal((128, 128))
            a = q @ q.T / 128 + np.eye(128)
            sc = 10.0 ** rng.uniform(-6, 6, 128)
            a = a * sc[:, None] * sc[None, :]
        else:
            u, _ = np.linalg.qr(rng.standard_normal((128, 128)))
            a = (u * np.logspace(0, -10, 128)) @ u.T
            a = (a + a.T) / 2
        ad = dev(a)
        inv = ops.zeros(128, 128)
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        ops.leaf_raw(ad, inv, info)
        assert int(info.item()) == 0
        chol = np.linalg.cholesky(a)
        full_l, full_i = host(ad), host(inv)
        got = np.tril(full_l)
        dscale = np.sqrt(np.diag(a))
        # against LAPACK, entry by entry relative to sqrt(a_ii) (|L_ij| <= sqrt(a_ii)): forward error grows with the condition number
        assert np.abs((got - chol) / dscale[:, None]).max() <= (1e-13 if kind == "scaled" else 1e-6)
        # L L^T reproduces the tile to working precision relative to its own scale (backward stability)
        assert np.abs((got @ got.T - a) / np.outer(dscale, dscale)).max() <= 5e-14
        # the inverse: inv L = I, relative to |inv| |L|
        gi = np.tril(full_i)
        den = np.abs(gi) @ np.abs(got)                       # lower triangular, like the product itself
        num = np.abs(gi @ got - np.eye(128))
        assert np.all(num[den == 0] == 0) and (num[den > 0] / den[den > 0]).max() <= 1e-13
        # the zeros above the diagonal of both tiles are part of the contract (later products treat diagonal tiles as full)
        assert np.all(np.triu(full_l, 1) == 0) and np.all(np.triu(full_i, 1) == 0)


def test_potrf_fp32(ops):
    rng = np.random.default_rng(5)
    n = 512
    a = spd(n, rng, 2.0)
    ad = dev(a, torch.float32)
    invd = ops.potrf_workspace(n, torch.float32)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.potrf(ad, invd, info)
    assert int(info.item()) == 0
    np.testing.assert_allclose(np.tril(host(ad)), np.linalg.cholesky(a), atol=2e-5)
    minv = ops.zeros(n, n, dtype=torch.float32)
    ops.trtri(ad, invd, minv)
    kinv = ops.zeros(n, n, dtype=torch.float32)
    ops.lauum(minv, kinv)
    np.testing.assert_allclose(np.tril(host(kinv)), np.tril(np.linalg.inv(a)), atol=2e-4)


# --------------------------------------------------------------------------- NLML pieces
@pytest.mark.parametrize("covs,d,n", [([orc.SE, orc.WN], 3, 333), ([orc.SE, orc.SE, orc.WN], 8, 333), ([orc.M52, orc.WN], 5, 333),
                                      ([orc.SE, orc.WN], 16, 333),
                                      ([orc.SE, orc.WN], 8, 1500),     # 24 tile rows: two strips of column tiles per row
                                      ([orc.SE, orc.WN], 40, 333),     # the d <= 64 instantiation (> 64 KB of LDS)
                                      ([orc.WN], 3, 200)])             # pure white noise: only the trace of W
def test_nlml_value_and_grad(ops, covs, d, n):
    from pygpr_amd._ops import pad_to

    x, y = orc.synth(n, d, seed=d)
    rng = np.random.default_rng(d)
    hp = np.concatenate([0.6 + 0.6 * rng.random(c.nhp(d)) if c.kind != "wn" else np.array([0.1]) for c in covs])
    npad = pad_to(n)
    spec = _spec(covs, d)
    hpd, xd = dev(hp), dev(x)
    k = ops.empty(npad, npad)
    ops.kernel_build(spec, hpd, xd, None, k, lower_only=True, jitter=1e-7)
    invd = ops.potrf_workspace(npad, torch.float64)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.potrf(k, invd, info)
    assert int(info.item()) == 0
    ypad = ops.zeros(npad)
    ypad[:n] = dev(y)
    alpha = ops.empty(npad)
    ops.potrs_vec(k, invd, ypad, alpha)
    out = ops.zeros(1 + hp.size)
    ops.nlml_value(k, ypad, alpha, n, out)
    minv = ops.zeros(npad, npad)
    ops.trtri(k, invd, minv)
    kinv = ops.zeros(npad, npad)
    ops.lauum(minv, kinv)
    work = ops.empty(ops.nlml_grad_worksize(n, hp.size))
    ops.nlml_grad(spec, hpd, xd, n, kinv, alpha, out[1:], work)
    loss_ref, grad_ref = orc.mle_loss_and_grad(covs, hp, x, y, "kinv", form="direct")
    got = host(out)
    np.testing.assert_allclose(got[0], loss_ref, rtol=1e-11)
    np.testing.assert_allclose(got[1:], grad_ref, rtol=1e-8, atol=1e-9 * np.abs(grad_ref).max())


def test_predict_mean_q(ops):
    from pygpr_amd._ops import pad_to

    n, d, m = 500, 4, 70
    covs = [orc.SE, orc.WN]
    x, y = orc.synth(n, d, seed=3)
    xp = np.random.default_rng(8).random((m, d))
    hp = np.array([1.0, 0.8, 1.1, 0.9, 1.2, 0.1])
    npad, mpad = pad_to(n), pad_to(m)
    spec = _spec(covs, d)
    hpd, xd, xpd = dev(hp), dev(x), dev(xp)
    k = ops.empty(npad, npad)
    ops.kernel_build(spec, hpd, xd, None, k, jitter=1e-7)
    invd = ops.potrf_workspace(npad, torch.float64)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.potrf(k, invd, info)
    ypad = ops.zeros(npad)
    ypad[:n] = dev(y)
    alpha = ops.empty(npad)
    ops.potrs_vec(k, invd, ypad, alpha)
    minv = ops.zeros(npad, npad)
    ops.trtri(k, invd, minv)
    ks = ops.empty(npad, mpad)
    ops.kernel_build(spec, hpd, xd, xpd, ks)
    mean, q = ops.empty(mpad), ops.empty(mpad)
    work = ops.empty((npad // 64) * mpad)
    kss = hp[0] ** 2 + hp[-1] ** 2
    ops.predict_mean_q(ks, minv, alpha, mean, q, kss, work)
    mu_ref, var_ref = orc.gp_predict(covs, hp, x, y, xp, "diag", form="direct")
    np.testing.assert_allclose(host(mean)[:m], mu_ref, atol=1e-10)
    np.testing.assert_allclose(host(q)[:m], var_ref, atol=1e-11)
    # the same from the cross-covariance stored test-point-major (the form the class surface uses)
    kt = ops.empty(mpad, npad)
    ops.kernel_build(spec, hpd, xpd, xd, kt)
    np.testing.assert_array_equal(host(kt)[:m, :n], host(ks)[:n, :m].T)
    mean2, q2 = ops.empty(mpad), ops.empty(mpad)
    ops.predict_mean_q_kt(kt, minv, alpha, mean2, q2, kss, work)
    np.testing.assert_allclose(host(mean2)[:m], mu_ref, atol=1e-10)
    np.testing.assert_allclose(host(q2)[:m], var_ref, atol=1e-11)
    # full covariance pieces: V = Minv Ks, C = Kss - V^T V
    v = ops.empty(npad, mpad)
    ops.trmm_lower(minv, ks, v)
    c = ops.zeros(mpad, mpad)
    c[:m, :m] = dev(orc.kernel(covs, hp, xp, form="direct"))
    ops.syrk_tn_sub(v, c)
    _, cov_ref = orc.gp_predict(covs, hp, x, y, xp, "full", form="direct")
    np.testing.assert_allclose(np.tril(host(c)[:m, :m]), np.tril(cov_ref), atol=1e-11)
    # the same pieces from the test-point-major cross-covariance (what the class surface uses): Vt = Kt Minv^T, and the rank-n update
    # of several experts' outputs in one launch (here: the same expert three times, one of them with a scaled Vt)
    vt = ops.empty(mpad, npad)
    ops.trmm_lower_kt(minv, kt, vt)
    np.testing.assert_allclose(host(vt), host(v).T, atol=1e-12)
    # ... and the products of several experts in one launch: inverses as a stack, or as a list of views at one stride
    minv3 = torch.stack([minv, 2.0 * minv, minv])
    kt3 = torch.stack([kt, kt, 0.5 * kt])
    for mm in (minv3, [minv3[0], minv3[1], minv3[2]]):
        vt3 = ops.empty(3, mpad, npad)
        ops.trmm_lower_kt(mm, kt3, vt3)
        assert np.array_equal(host(vt3[0]), host(vt))
        np.testing.assert_allclose(host(vt3[1]), 2.0 * host(vt), rtol=1e-15)
        np.testing.assert_allclose(host(vt3[2]), 0.5 * host(vt), rtol=1e-15)
    vt_all = torch.stack([vt, 0.5 * vt, vt])
    c_all = ops.zeros(3, mpad, mpad)
    c_all[:, :m, :m] = dev(orc.kernel(covs, hp, xp, form="direct"))
    c_keep = host(c_all[1]).copy()
    ops.syrk_nt_sub_batched(vt_all, c_all)
    for e in (0, 2):
        np.testing.assert_allclose(np.tril(host(c_all[e])[:m, :m]), np.tril(cov_ref), atol=1e-11)
    assert np.array_equal(host(c_all[0]), host(c_all[2]))
    np.testing.assert_allclose(np.tril(host(c_all[1])), np.tril(c_keep - 0.25 * host(v).T @ host(v)), atol=1e-11)
    assert np.array_equal(np.triu(host(c_all[1]), 128), np.triu(c_keep, 128))          # tiles above the diagonal are not touched


def test_grbcm_terms(ops):
    rng = np.random.default_rng(6)
    m = 300
    mc, vc, vg, mg = rng.standard_normal(m), 0.1 + rng.random(m), 0.2 + rng.random(m), rng.standard_normal(m)
    out = ops.zeros(3, m)
    beta, prec = ops.empty(m), ops.empty(m)
    ops.grbcm_local_terms(dev(mc), dev(vc), dev(vg), True, False, out, beta, prec)
    np.testing.assert_allclose(host(beta), 1.0)
    np.testing.assert_allclose(host(prec), 1.0 / vc, rtol=1e-14)
    ops.grbcm_local_terms(dev(mc * 0.5), dev(vc * 1.3), dev(vg), False, True, out)
    ref = orc.grbcm_terms(mc, vc, vg, True) + orc.grbcm_terms(mc * 0.5, vc * 1.3, vg, False)
    np.testing.assert_allclose(host(out), ref, rtol=1e-13)
    mean, var = ops.empty(m), ops.empty(m)
    ops.grbcm_finish(out, dev(mg), dev(vg), mean, var)
    mu_ref, var_ref = orc.grbcm_finish(ref, mg, vg)
    np.testing.assert_allclose(host(mean), mu_ref, rtol=1e-12)
    np.testing.assert_allclose(host(var), var_ref, rtol=1e-12)


# ---- flag-coupled chain (chainstep.hip): resident leaf + rows kernels handing over through device flags ------------------------
CS_PANEL = 384      # width of a coupled panel (PG_CS_PANEL; round 3: 512 -> 384) -- the coupled region is the last 8192 rows


def coupled_count(n):
    """Outer panels that run on the coupled chain for an n x n factorisation with the default schedule."""
    first = 0
    nbo = 512 if n <= 10240 else (1024 if n <= 18432 else 2048)
    while n - first > 8192:
        first += nbo
    return -(-(n - first) // CS_PANEL)


def _potrf_on_compute_stream(ops, a, dtype=torch.float64):
    """pg_potrf from the caller's current stream (the legacy default stream here: the rows kernels run on the handle's own
    rows stream, so the caller's stream does not matter)."""
    ad = dev(a, dtype)
    n = a.shape[0]
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    invd = ops.potrf_workspace(n, dtype)
    ops.potrf(ad, invd, info)
    coupled = ops.last_coupled_panels()
    torch.cuda.synchronize()
    return ad, invd, int(info.item()), coupled


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1536, 2816, 4096, 6144])
def test_coupled_chain_factor_matches_lapack(ops, n):
    """Every panel of these sizes runs on the coupled chain (n <= 8192): factor and diagonal-block inverses against LAPACK."""
    rng = np.random.default_rng(100 + n)
    a = spd(n, rng)
    ad, invd, info, coupled = _potrf_on_compute_stream(ops, a)
    assert info == 0
    assert coupled == coupled_count(n), "the coupled chain did not run"
    chol = np.linalg.cholesky(a)
    np.testing.assert_allclose(np.tril(host(ad)), chol, atol=1e-11)
    blocks = host(invd[: n * 128]).reshape(n // 128, 128, 128)
    for b in (0, n // 256, n // 128 - 1):
        d = chol[b * 128:(b + 1) * 128, b * 128:(b + 1) * 128]
        np.testing.assert_allclose(blocks[b], np.linalg.inv(d), atol=1e-10)


@pytest.mark.gpu
def test_coupled_chain_tail_of_a_larger_matrix(ops):
    """n = 10240: the first panels run the classic chain, the last 8192 rows the coupled one; same factor as the single-stream
    schedule to rounding."""
    n = 10240
    rng = np.random.default_rng(7)
    a = spd(n, rng)
    ad, _, info, coupled = _potrf_on_compute_stream(ops, a)
    assert info == 0 and coupled == coupled_count(n)
    ops.set_lookahead(0)
    try:
        ref = dev(a)
        info2 = torch.zeros(1, dtype=torch.int32, device="cuda")
        ops.potrf(ref, ops.potrf_workspace(n, torch.float64), info2)
        assert int(info2.item()) == 0 and ops.last_coupled_panels() == 0
    finally:
        ops.set_lookahead(1)
    np.testing.assert_allclose(np.tril(host(ad)), np.tril(host(ref)), rtol=0, atol=5e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("bad", [0, 700, 2047, 3000])
def test_coupled_chain_reports_the_first_bad_pivot_and_ends(ops, bad):
    """A non-positive pivot inside the coupled region: info = column + 1 (LAPACK), and every resident kernel still gets its
    flags (the call returns instead of spinning)."""
    n = 3072
    rng = np.random.default_rng(5)
    a = spd(n, rng)
    a[bad, bad] = -1.0
    _, _, info, coupled = _potrf_on_compute_stream(ops, a)
    assert coupled > 0
    assert info == bad + 1


@pytest.mark.gpu
def test_coupled_chain_fp32(ops):
    n = 4096
    rng = np.random.default_rng(11)
    a = spd(n, rng, cond_shift=4.0)
    ad, _, info, coupled = _potrf_on_compute_stream(ops, a, torch.float32)
    assert info == 0 and coupled == coupled_count(n)
    l = np.tril(host(ad).astype(np.float64))
    assert np.abs(l @ l.T - a).max() / np.abs(a).max() < 5e-5


@pytest.mark.gpu
def test_coupled_chain_fused_inverse(ops):
    """pg_potrf_trtri at n = 4096: coupled chain + the triangular inverse behind it."""
    n = 4096
    rng = np.random.default_rng(13)
    a = spd(n, rng)
    ad = dev(a)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    minv = ops.zeros(n, n)
    ops.potrf_trtri(ad, ops.potrf_workspace(n, torch.float64), info, minv)
    coupled = ops.last_coupled_panels()
    torch.cuda.synchronize()
    assert int(info.item()) == 0 and coupled == coupled_count(n)
    chol = np.linalg.cholesky(a)
    np.testing.assert_allclose(np.tril(host(minv)), np.linalg.inv(chol), atol=1e-9)


@pytest.mark.gpu
def test_coupled_chain_from_a_side_stream_caller(ops):
    """The caller on a non-default torch stream: same schedule, the result is ordered on that stream."""
    n = 3072
    a = spd(n, np.random.default_rng(3))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ad = dev(a)
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        ops.potrf(ad, ops.potrf_workspace(n, torch.float64), info)
        coupled = ops.last_coupled_panels()
        lower = torch.tril(ad).cpu().numpy()          # on the side stream, behind the factorisation
    assert int(info.item()) == 0 and coupled == coupled_count(n)
    np.testing.assert_allclose(lower, np.linalg.cholesky(a), atol=1e-11)


@pytest.mark.gpu
def test_coupled_chain_switch_and_probe(ops):
    """pg_create's probe found concurrent queues on a plain run (the chain is on); switching it off gives the classic chain
    and the same factor to rounding."""
    assert ops.coupled_chain() == 1
    n = 3072
    a = spd(n, np.random.default_rng(17))
    outs = []
    try:
        for on in (1, 0):
            ops.set_coupled_chain(on)
            ad, _, info, coupled = _potrf_on_compute_stream(ops, a)
            assert info == 0 and (coupled == coupled_count(n) if on else coupled == 0)
            outs.append(np.tril(host(ad)))
    finally:
        ops.set_coupled_chain(1)
    np.testing.assert_allclose(outs[0], outs[1], rtol=0, atol=2e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("n,with_inv", [(1000, False), (3072, True), (10240, False)])
def test_build_folded_into_the_factorisation(ops, n, with_inv):
    """pg_build_potrf_trtri == pg_kernel_build(lower_only) + pg_potrf[_trtri]: same matrix, same schedule, the build of the
    columns right of the first panel merely runs beside that panel's chain -- bit-identical factor (and inverse)."""
    from pygpr_amd._ops import pad_to

    d = 5
    covs = [orc.SE, orc.WN]
    x, _ = orc.synth(n, d, seed=n)
    hp = np.array([1.0, 0.9, 1.1, 0.8, 1.2, 1.0, 0.1])
    npad = pad_to(n)
    spec = _spec(covs, d)
    hpd, xd = dev(hp), dev(x)
    outs = []
    for folded in (False, True):
        a = ops.empty(npad, npad)
        a.fill_(float("nan"))
        invd = ops.potrf_workspace(npad, torch.float64)
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        minv = ops.zeros(npad, npad) if with_inv else None
        if folded:
            ops.build_factor(spec, hpd, xd, a, invd, info, minv, jitter=1e-7)
        else:
            ops.kernel_build(spec, hpd, xd, None, a, lower_only=True, jitter=1e-7)
            (ops.potrf_trtri(a, invd, info, minv) if with_inv else ops.potrf(a, invd, info))
        assert int(info.item()) == 0
        outs.append((np.tril(host(a)), np.tril(host(minv)) if with_inv else None))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    if with_inv:
        np.testing.assert_array_equal(outs[0][1], outs[1][1])
    k = orc.kernel(covs, hp, x, form="direct") + 1e-7 * np.eye(n)
    np.testing.assert_allclose(outs[1][0][:n, :n], np.linalg.cholesky(k), atol=1e-10)


# ---- the coupled chain's time-out ends in a correct factor (tc.cholesky, PyGPR/gpr.py:69, never fails on a PD matrix) ----------
@pytest.fixture
def forced_timeout(ops):
    """Every wait of the coupled chain expires at once (pg_set_spin_budget(h, -1)): the deterministic stand-in for an environment
    whose queues do not run concurrently.  The chain is re-armed afterwards whatever the test did."""
    assert ops.coupled_chain() == 1
    ops.set_spin_budget(-1)
    yield
    ops.set_spin_budget(0)
    ops.set_coupled_chain(1)
    assert ops.coupled_chain() == 1


@pytest.mark.gpu
def test_coupled_chain_timeout_reports_and_switches_to_the_classic_chain(ops, forced_timeout):
    """C ABI, asynchronous entry point: info = -1, the handle has switched itself off the coupled chain by the next entry point
    (pg_coupled_chain() == 0, pg_chain_timeouts() counted it), and repeating the call gives LAPACK's factor."""
    n = 3072
    a = spd(n, np.random.default_rng(23))
    before = ops.chain_timeouts()
    _, _, info, coupled = _potrf_on_compute_stream(ops, a)
    assert coupled == coupled_count(n) and info == -1
    assert ops.chain_timeouts() == before + 1 and ops.coupled_chain() == 0
    ad, _, info, coupled = _potrf_on_compute_stream(ops, a)
    assert info == 0 and coupled == 0
    np.testing.assert_allclose(np.tril(host(ad)), np.linalg.cholesky(a), atol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("with_inv", [False, True])
def test_build_potrf_trtri_checked_falls_back_inside_the_call(ops, forced_timeout, with_inv):
    """pg_build_potrf_trtri_checked: the time-out is repaired INSIDE the call (rebuild + classic chain); the caller sees info = 0,
    LAPACK's factor (1e-11) and inverse, pg_coupled_chain() == 0 afterwards, and the next call is as fast as the classic chain."""
    import time

    from pygpr_amd._ops import pad_to

    n, d = 3000, 5
    covs = [orc.SE, orc.WN]
    x, _ = orc.synth(n, d, seed=31)
    hp = np.array([1.0, 0.9, 1.1, 0.8, 1.2, 1.0, 0.1])
    npad = pad_to(n)
    spec, hpd, xd = _spec(covs, d), dev(hp), dev(x)
    a = ops.empty(npad, npad)
    invd = ops.potrf_workspace(npad, torch.float64)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    minv = ops.zeros(npad, npad) if with_inv else None
    before = ops.chain_timeouts()
    assert ops.build_factor_checked(spec, hpd, xd, a, invd, info, minv, jitter=1e-7) == 0
    assert ops.chain_timeouts() == before + 1 and ops.coupled_chain() == 0 and ops.last_coupled_panels() == 0
    k = orc.kernel(covs, hp, x, form="direct") + 1e-7 * np.eye(n)
    chol = np.linalg.cholesky(k)
    np.testing.assert_allclose(np.tril(host(a))[:n, :n], chol, atol=1e-11)
    if with_inv:
        np.testing.assert_allclose(np.tril(host(minv))[:n, :n], np.linalg.inv(chol), atol=1e-8)
    # next call: no time-out left to wait for (a 2 s budget would show), classic chain
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert ops.build_factor_checked(spec, hpd, xd, a, invd, info, minv, jitter=1e-7) == 0
    assert time.perf_counter() - t0 < 0.25 and ops.last_coupled_panels() == 0


@pytest.mark.gpu
def test_batched_coupled_chain_timeout_falls_back(ops, forced_timeout):
    """Experts together on the coupled chain (3 x 2304 points) with every wait forced to expire: every expert reports info = -1, the
    handle switches itself to the classic chain, and the repeated call gives LAPACK's factors -- what Exact_GP.update / MLE do at their
    one synchronisation point (gpr._checked)."""
    import os
    if os.environ.get("PG_CS_BATCHED") == "0":
        pytest.skip("the batch stays on the classic chain with PG_CS_BATCHED=0")
    nexp, n, d = 3, 2304, 3
    covs = [orc.SE, orc.WN]
    rng = np.random.default_rng(77)
    x = rng.random((nexp, n, d))
    hp = np.tile(np.array([1.0, 0.9, 1.1, 0.8, 0.2]), (nexp, 1))
    spec = _spec(covs, d)
    a = ops.empty(nexp, n, n)
    invd = ops.empty(nexp, ops.potrf_worksize(n, torch.float64))
    info = torch.zeros(nexp, dtype=torch.int32, device="cuda")
    before = ops.chain_timeouts()
    ops.build_factor_batched(spec, dev(hp), dev(x), n * d, a, invd, info, None, jitter=1e-7)
    assert ops.last_coupled_panels() == coupled_count(n)
    torch.cuda.synchronize()
    assert info.tolist() == [-1] * nexp
    assert ops.chain_timeouts() == before + 1 and ops.coupled_chain() == 0
    ops.build_factor_batched(spec, dev(hp), dev(x), n * d, a, invd, info, None, jitter=1e-7)
    assert info.tolist() == [0] * nexp and ops.last_coupled_panels() == 0
    for e in range(nexp):
        k = orc.kernel(covs, hp[e], x[e], form="direct") + 1e-7 * np.eye(n)
        np.testing.assert_allclose(np.tril(host(a[e])), np.linalg.cholesky(k), atol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("nexp,n,d,shared_x", [(4, 1500, 8, False), (10, 100, 3, False), (3, 700, 16, True)])
def test_batched_gradient_path_matches_the_single_expert_calls(ops, nexp, n, d, shared_x):
    """Round 4: pg_alpha_nlml_batched / pg_lauum_batched / pg_nlml_grad_batched after pg_build_potrf_trtri_batched -- the gradient
    path of loss.py:92-128 for a model with a leading expert dimension -- against the single-expert entry points on each expert's
    slice (same kernels, one grid dimension wider: bit for bit) and against the oracle (NLML 1e-10, gradient 1e-8).  d = 8: the fast
    contraction; d = 16: the general one; shared_x: batched hyper-parameters on one set of points (stride 0)."""
    from pygpr_amd._ops import pad_to

    covs = [orc.SE, orc.WN]
    rng = np.random.default_rng(7 * nexp + n)
    x = rng.random((1 if shared_x else nexp, n, d))
    y = np.sin(x.sum(-1) * 3.0).repeat(nexp if shared_x else 1, axis=0) + 0.05 * rng.standard_normal((nexp, n))
    hp = np.concatenate([0.8 + 0.4 * rng.random((nexp, 1 + d)), np.full((nexp, 1), 0.15)], axis=1)
    npad = pad_to(n)
    spec = _spec(covs, d)
    hpd, xd = dev(hp), dev(x)
    ypad = ops.zeros(nexp, npad)
    ypad[:, :n] = dev(y)
    a, m = ops.empty(nexp, npad, npad), ops.empty(nexp, npad, npad)
    invd = ops.empty(nexp, ops.potrf_worksize(npad, torch.float64))
    info = torch.ones(nexp, dtype=torch.int32, device="cuda")
    x_stride = 0 if shared_x else xd.stride(0)
    ops.build_factor_batched(spec, hpd, xd, x_stride, a, invd, info, m, jitter=1e-7)
    assert info.tolist() == [0] * nexp
    u, alpha = ops.empty(nexp, npad), ops.empty(nexp, npad)
    vwork = ops.empty(nexp, (npad // 256) * npad)
    outs = ops.zeros(nexp, 2 + d + 1)
    ops.alpha_nlml_batched(m, ypad, u, alpha, vwork, n, outs)
    kinv = ops.empty(nexp, npad, npad)
    ops.lauum_batched(m, kinv)
    gwork = ops.empty(nexp * ops.nlml_grad_worksize(n, hp.shape[1]))
    ops.nlml_grad_batched(spec, hpd, xd, x_stride, n, kinv, alpha, outs[:, 1:], gwork)
    got = host(outs)
    for e in range(nexp):
        xe = xd[0 if shared_x else e]
        a1, u1, w1 = ops.empty(npad), ops.empty(npad), ops.empty((npad // 256) * npad)
        ops.trmv(m[e], ypad[e], u1, 0)
        ops.trmv(m[e], u1, a1, 1, w1)
        assert torch.equal(a1, alpha[e])
        k1 = ops.empty(npad, npad)
        ops.lauum(m[e], k1)
        assert torch.equal(torch.tril(k1), torch.tril(kinv[e]))
        g1 = ops.zeros(hp.shape[1])
        ops.nlml_grad(spec, hpd[e], xe, n, k1, a1, g1, ops.empty(ops.nlml_grad_worksize(n, hp.shape[1])))
        assert torch.equal(g1, outs[e, 1:])
        loss_ref, grad_ref = orc.mle_loss_and_grad(covs, hp[e], x[0 if shared_x else e], y[e], "kinv", form="direct")
        np.testing.assert_allclose(got[e, 0], loss_ref, rtol=1e-10)
        np.testing.assert_allclose(got[e, 1:], grad_ref, rtol=1e-8, atol=1e-9 * np.abs(grad_ref).max())


@pytest.mark.gpu
def test_gemm_many_small_experts_exceed_one_grid(ops):
    """grid.y = pairs x experts is limited to 65535: 300 experts of 512 points make the triangular inverse's doubling pass a launch of
    1 x 300 and the batched products chunk by whole experts beyond the limit -- exercised here by forcing tiny chunks is not possible
    from outside, so the test runs the largest cheap case (nexp = 600, n_pad = 256: one pair each) and checks every expert."""
    nexp, n, d = 600, 200, 2
    rng = np.random.default_rng(5)
    x = rng.random((nexp, n, d))
    hp = np.tile(np.array([1.0, 0.9, 1.1, 0.2]), (nexp, 1))
    a, m = ops.empty(nexp, 256, 256), ops.empty(nexp, 256, 256)
    invd = ops.empty(nexp, ops.potrf_worksize(256, torch.float64))
    info = torch.ones(nexp, dtype=torch.int32, device="cuda")
    ops.build_factor_batched(_spec([orc.SE, orc.WN], d), dev(hp), dev(x), n * d, a, invd, info, m, jitter=1e-7)
    assert info.tolist() == [0] * nexp
    for e in (0, 299, 599):
        k = orc.kernel([orc.SE, orc.WN], hp[e], x[e], form="direct") + 1e-7 * np.eye(n)
        chol = np.linalg.cholesky(k)
        np.testing.assert_allclose(np.tril(host(a[e]))[:n, :n], chol, atol=1e-11)
        np.testing.assert_allclose(np.tril(host(m[e]))[:n, :n], np.linalg.inv(chol), atol=1e-9)


@pytest.mark.gpu
def test_a_timeout_is_temporary_the_handle_rearms_itself(ops):
    """Round 4: a time-out switches the handle to the classic chain only for pg_set_rearm_after(h, K) further factorisations; the next
    one probes the queues again and takes the coupled chain back by itself (round 3: the downgrade lasted for the life of the handle,
    profiles/r03_bench_n2_gloo_rehearsal.json).  Forced expiry -> fall-back, K = 3 clean calls on the classic chain, automatic
    re-arm, pg_last_coupled_panels() > 0 again, LAPACK's factor throughout.  The wait budget is back on its default (scaled to the
    call) while the classic calls run."""
    n = 3072
    a = spd(n, np.random.default_rng(29))
    chol = np.linalg.cholesky(a)
    assert ops.coupled_chain() == 1
    rearms, tmos = ops.chain_rearms(), ops.chain_timeouts()
    ops.set_rearm_after(3)
    try:
        ops.set_spin_budget(-1)
        _, _, info, coupled = _potrf_on_compute_stream(ops, a)
        assert info == -1 and coupled == coupled_count(n)
        ops.set_spin_budget(0)
        assert ops.chain_timeouts() == tmos + 1 and ops.coupled_chain() == 0
        for _ in range(3):
            ad, _, info, coupled = _potrf_on_compute_stream(ops, a)
            assert info == 0 and coupled == 0 and ops.coupled_chain() == 0 and ops.chain_rearms() == rearms
            np.testing.assert_allclose(np.tril(host(ad)), chol, atol=1e-11)
        ad, _, info, coupled = _potrf_on_compute_stream(ops, a)      # the K+1-th call re-arms before it factorises
        assert info == 0 and coupled == coupled_count(n) and ops.coupled_chain() == 1 and ops.chain_rearms() == rearms + 1
        np.testing.assert_allclose(np.tril(host(ad)), chol, atol=1e-11)
        # a time-out that FOLLOWS a re-arm doubles the distance (a GPU shared for good with another process' resident kernels must not
        # cost a wait budget every K calls): six classic calls now, the seventh re-arms
        ops.set_spin_budget(-1)
        _, _, info, coupled = _potrf_on_compute_stream(ops, a)
        assert info == -1 and coupled == coupled_count(n)
        ops.set_spin_budget(0)
        assert ops.chain_timeouts() == tmos + 2 and ops.coupled_chain() == 0
        for _ in range(6):
            _, _, info, coupled = _potrf_on_compute_stream(ops, a)
            assert info == 0 and coupled == 0 and ops.chain_rearms() == rearms + 1
        ad, _, info, coupled = _potrf_on_compute_stream(ops, a)
        assert info == 0 and coupled == coupled_count(n) and ops.chain_rearms() == rearms + 2
        np.testing.assert_allclose(np.tril(host(ad)), chol, atol=1e-11)
        # a caller's explicit "off" is not a time-out: no automatic re-arm
        ops.set_coupled_chain(0)
        for _ in range(5):
            _, _, info, coupled = _potrf_on_compute_stream(ops, a)
            assert info == 0 and coupled == 0
        assert ops.coupled_chain() == 0 and ops.chain_rearms() == rearms + 2
    finally:
        ops.set_rearm_after(8)
        ops.set_spin_budget(0)
        ops.set_coupled_chain(1)
    assert ops.coupled_chain() == 1


@pytest.mark.gpu
def test_rearm_counts_batched_calls_and_the_backoff_decays(ops):
    """Round 5 (ADVICE): (i) pg_build_potrf_trtri_batched counts towards the automatic re-arm like the single-matrix calls -- a
    batched-only workload (MLE on a batched model) gets the coupled chain back after K calls; (ii) the back-off does not ratchet: a
    re-armed chain that ran cleanly for a whole back-off distance has the short distance again, so the NEXT time-out doubles K, not 2K."""
    n, nexp = 2048, 2
    rng = np.random.default_rng(31)
    a = spd(n, rng)
    chol = np.linalg.cholesky(a)
    assert ops.coupled_chain() == 1
    rearms = ops.chain_rearms()

    def batched():
        a_all = dev(np.stack([a, a]))
        invd_all = ops.empty(nexp, ops.potrf_worksize(n, torch.float64))
        info_all = torch.zeros(nexp, dtype=torch.int32, device="cuda")
        ops.potrf_trtri_batched(a_all, invd_all, info_all, None)
        coupled = ops.last_coupled_panels()
        torch.cuda.synchronize()
        assert not info_all.any().item()
        np.testing.assert_allclose(np.tril(host(a_all[1])), chol, atol=1e-11)
        return coupled

    def time_out():
        ops.set_spin_budget(-1)
        _, _, info, _ = _potrf_on_compute_stream(ops, a)
        ops.set_spin_budget(0)
        before = ops.chain_timeouts()        # (an entry point that polls the pinned time-out word)
        assert info == -1 and before >= 1 and ops.coupled_chain() == 0

    ops.set_rearm_after(2)
    try:
        time_out()
        for _ in range(2):
            assert batched() == 0 and ops.chain_rearms() == rearms
        assert batched() > 0 and ops.coupled_chain() == 1 and ops.chain_rearms() == rearms + 1     # the third batched call re-arms
        time_out()                                   # follows a re-arm: distance 4
        for _ in range(4):
            assert _potrf_on_compute_stream(ops, a)[3] == 0
        assert _potrf_on_compute_stream(ops, a)[3] > 0 and ops.chain_rearms() == rearms + 2
        for _ in range(5):                           # more clean coupled calls than the back-off distance: it decays to 2
            assert _potrf_on_compute_stream(ops, a)[3] > 0
        time_out()                                   # doubles 2, not 4
        for _ in range(4):
            assert _potrf_on_compute_stream(ops, a)[3] == 0 and ops.chain_rearms() == rearms + 2
        assert _potrf_on_compute_stream(ops, a)[3] > 0 and ops.chain_rearms() == rearms + 3
    finally:
        ops.set_rearm_after(8)
        ops.set_spin_budget(0)
        ops.set_coupled_chain(1)
    assert ops.coupled_chain() == 1


@pytest.mark.gpu
def test_wait_budget_is_scaled_to_the_call(ops):
    """The default budget of one wait is 20x the call's classic-chain estimate, at least 50 ms -- not round 3's flat 2 s (70x a whole
    N = 16384 factorisation; in a lock-step all-reduce one rank's stall is every rank's).  The figures, and a REAL wait on a flag nobody
    sets with the budget of n = 3072: it gives up after about 50 ms."""
    assert ops.wait_budget_us(2048) == 50_000 and 50_000 <= ops.wait_budget_us(4096) < 60_000
    assert 150_000 < ops.wait_budget_us(8192) < 250_000 and 800_000 < ops.wait_budget_us(16384) < 1_300_000
    try:
        ops.set_spin_budget(1234)
        assert ops.wait_budget_us(16384) == 1234
        ops.set_spin_budget(-1)
        assert ops.wait_budget_us(16384) == -1
        ms, came = ops.spin_probe(3072)
        assert not came and ms < 1.0
    finally:
        ops.set_spin_budget(0)
    before = ops.chain_timeouts()
    ms, came = ops.spin_probe(3072)
    assert not came and 45.0 <= ms <= 80.0
    assert ops.chain_timeouts() == before and ops.coupled_chain() == 1      # the probe does not touch the handle


# ---- batched experts: one call, every launch covers all of them (PyGPR/gpr.py:65-74 factorises a batch in one tc.cholesky) ------
@pytest.mark.gpu
@pytest.mark.parametrize("nexp,n,with_inv", [(8, 4096, True), (3, 1000, False), (10, 100, True), (3, 2500, True), (5, 4000, False)])
def test_batched_factorisation(ops, nexp, n, with_inv):
    """pg_build_potrf_trtri_batched against LAPACK per expert (factor 1e-11, inverse 1e-9), and bit for bit against the
    single-expert call on the same chain -- classic, or coupled where the batch takes it (the batch only widens the grids);
    expert 1 gets different hyper-parameters, and a
    NaN in ONE expert's points fails that expert alone (info[e] > 0, the others' factors untouched)."""
    from pygpr_amd._ops import pad_to

    d = 4
    covs = [orc.SE, orc.WN]
    rng = np.random.default_rng(1000 + n)
    x = rng.random((nexp, n, d))
    hp = np.tile(np.array([1.0, 0.9, 1.1, 0.8, 1.2, 0.1]), (nexp, 1))
    hp[1] = [1.3, 0.5, 0.7, 1.4, 0.6, 0.2]
    npad = pad_to(n)
    spec = _spec(covs, d)
    hpd, xd = dev(hp), dev(x)
    a = ops.empty(nexp, npad, npad)
    invd = ops.empty(nexp, ops.potrf_worksize(npad, torch.float64))
    info = torch.ones(nexp, dtype=torch.int32, device="cuda")
    minv = ops.zeros(nexp, npad, npad) if with_inv else None
    ops.build_factor_batched(spec, hpd, xd, xd.stride(0), a, invd, info, minv, jitter=1e-7)
    # round 4: a batch that is still latency-bound (experts of >= 2048 points, <= 24576 rows in all: the 3 x 2500 and 5 x 4000 cases)
    # takes the coupled chain -- one leaf workgroup and one grid row of rows workgroups per expert
    import os
    mode = os.environ.get("PG_CS_BATCHED", "1")          # 0: never, 1: the rule below, 2: whenever the look-ahead has three panels
    rule = (npad >= 2048 and nexp * npad <= 24576) if mode == "1" else (mode == "2" and npad >= 1024 and nexp <= 24)
    want_coupled = coupled_count(npad) if rule else 0
    assert info.tolist() == [0] * nexp and ops.last_coupled_panels() == want_coupled
    la, lm = host(a), (host(minv) if with_inv else None)
    for e in sorted({0, 1, nexp - 1}):
        k = orc.kernel(covs, hp[e], x[e], form="direct") + 1e-7 * np.eye(n)
        chol = np.linalg.cholesky(k)
        np.testing.assert_allclose(np.tril(la[e])[:n, :n], chol, atol=1e-11)
        if with_inv:
            np.testing.assert_allclose(np.tril(lm[e])[:n, :n], np.linalg.inv(chol), atol=1e-9)
    # the same expert alone on the same chain (coupled when the batch was): identical bits
    if not want_coupled:
        ops.set_coupled_chain(0)
    try:
        for e in (1, nexp - 1):
            a1 = ops.empty(npad, npad)
            i1 = torch.zeros(1, dtype=torch.int32, device="cuda")
            m1 = ops.zeros(npad, npad) if with_inv else None
            ops.build_factor(spec, hpd[e], xd[e], a1, ops.potrf_workspace(npad, torch.float64), i1, m1, jitter=1e-7)
            assert int(i1.item()) == 0
            np.testing.assert_array_equal(np.tril(host(a1)), np.tril(la[e]))
            if with_inv:
                np.testing.assert_array_equal(np.tril(host(m1)), np.tril(lm[e]))
    finally:
        ops.set_coupled_chain(1)
    # one bad expert fails alone
    xb = x.copy()
    xb[2 % nexp, 5, 0] = np.nan
    a2 = ops.empty(nexp, npad, npad)
    ops.build_factor_batched(spec, hpd, dev(xb), xd.stride(0), a2, invd, info, None, jitter=1e-7)
    got = info.tolist()
    assert got[2 % nexp] > 0 and all(v == 0 for i, v in enumerate(got) if i != 2 % nexp)
    np.testing.assert_array_equal(np.tril(host(a2[0])), np.tril(la[0]))
    # shared points (x_stride = 0): experts differ by their hyper-parameters only
    ops.build_factor_batched(spec, hpd, xd[:1].contiguous(), 0, a2, invd, info, None, jitter=1e-7)
    assert info.tolist() == [0] * nexp
    k = orc.kernel(covs, hp[1], x[0], form="direct") + 1e-7 * np.eye(n)
    np.testing.assert_allclose(np.tril(host(a2[1]))[:n, :n], np.linalg.cholesky(k), atol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("n,nrhs", [(768, 128), (2304, 384)])
def test_potrs_matrix_rhs_and_trsm(ops, n, nrhs):
    """pg_potrs / pg_trsm_lower against scipy's cho_solve / solve_triangular (tc.cholesky_solve with a matrix right-hand side,
    PyGPR/gpr.py:100,112, loss.py:116), with L^-1 formed inside the call and handed in."""
    import scipy.linalg as sla

    rng = np.random.default_rng(n)
    a = spd(n, rng)
    b = rng.standard_normal((n, nrhs))
    ad, bd = dev(a), dev(b)
    invd = ops.potrf_workspace(n, torch.float64)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.potrf(ad, invd, info)
    assert int(info.item()) == 0
    chol = np.linalg.cholesky(a)
    x_ref = sla.cho_solve((chol, True), b)
    v_ref = sla.solve_triangular(chol, b, lower=True)
    np.testing.assert_allclose(host(ops.potrs(ad, invd, bd)), x_ref, rtol=0, atol=1e-10 * np.abs(x_ref).max())
    np.testing.assert_allclose(host(ops.potrs(ad, invd, bd, triangular_only=True)), v_ref, rtol=0, atol=1e-11 * np.abs(v_ref).max())
    minv = ops.zeros(n, n)
    ops.trtri(ad, invd, minv)
    np.testing.assert_allclose(host(ops.potrs(None, None, bd, minv=minv)), x_ref, rtol=0, atol=1e-10 * np.abs(x_ref).max())


def test_batched_prediction_entry_points_match_the_one_expert_calls(ops):
    """pg_kernel_build_batched, pg_predict_mean_q_kt_batched, pg_grbcm_local_terms_batched (round 5): per expert the numbers of the
    one-expert entry points BIT FOR BIT (same kernels, the expert is one more grid dimension) -- cross builds with shared and with
    per-expert test points, shared and per-expert training points, means only and means + variances, fp64 and fp32."""
    from pygpr_amd._ops import make_spec

    rng = np.random.default_rng(21)
    nexp, n, m, d = 3, 300, 200, 5
    npad, mpad = 512, 256
    spec = make_spec([0], [0], [d + 1])
    for dtype in (torch.float64, torch.float32):
        x = rng.random((nexp, n, d))
        xp = rng.random((nexp, m, d))
        hp = np.stack([np.concatenate([[1.0 + 0.1 * e], 0.5 + rng.random(d), [0.1 + 0.05 * e]]) for e in range(nexp)])
        xd, xpd, hpd = dev(x, dtype), dev(xp, dtype), dev(hp)
        for shared_xp in (True, False):
            for shared_x in (False, True):
                kt_all = ops.empty(nexp, mpad, npad, dtype=dtype)
                xr = xpd[0] if shared_xp else xpd
                xc = xd[:1] if shared_x else xd
                ops.kernel_build_batched(spec, hpd, xr, xc, kt_all)
                for e in range(nexp):
                    kt = ops.empty(mpad, npad, dtype=dtype)
                    ops.kernel_build(spec, hpd[e], xpd[0 if shared_xp else e], xd[0 if shared_x else e], kt)
                    assert torch.equal(kt, kt_all[e]), (dtype, shared_xp, shared_x, e)
        # symmetric batched build (lower-only with jitter) against the one-expert call
        k_all = ops.zeros(nexp, npad, npad, dtype=dtype)
        ops.kernel_build_batched(spec, hpd, xd, None, k_all, lower_only=True, jitter=1e-7)
        for e in range(nexp):
            k1 = ops.zeros(npad, npad, dtype=dtype)
            ops.kernel_build(spec, hpd[e], xd[e], None, k1, lower_only=True, jitter=1e-7)
            assert torch.equal(k1, k_all[e])
        # factors, inverses, weights of the three experts (one-expert calls), then the batched prediction against the one-expert one
        minv_all = ops.zeros(nexp, npad, npad, dtype=dtype)
        alpha_all = ops.zeros(nexp, npad, dtype=dtype)
        for e in range(nexp):
            a = ops.empty(npad, npad, dtype=dtype)
            invd = ops.potrf_workspace(npad, dtype)
            info = torch.zeros(1, dtype=torch.int32, device="cuda")
            ops.build_factor(spec, hpd[e], xd[e], a, invd, info, minv_all[e])
            assert int(info.item()) == 0
            alpha_all[e, :n] = dev(rng.standard_normal(n), dtype)
        kt_all = ops.empty(nexp, mpad, npad, dtype=dtype)
        ops.kernel_build_batched(spec, hpd, xpd, xd, kt_all)
        work_all = ops.empty(nexp, (npad // 64) * mpad, dtype=dtype)
        for want_var in (True, False):
            mean_all, var_all = ops.empty(nexp, mpad, dtype=dtype), (ops.empty(nexp, mpad, dtype=dtype) if want_var else None)
            ops.predict_mean_q_kt_batched(kt_all, minv_all if want_var else None, alpha_all, mean_all, var_all, spec, hpd, work_all)
            for e in range(nexp):
                mean1, var1 = ops.empty(mpad, dtype=dtype), (ops.empty(mpad, dtype=dtype) if want_var else None)
                kss = float(hp[e, 0] ** 2 + hp[e, -1] ** 2)
                ops.predict_mean_q_kt(kt_all[e], minv_all[e] if want_var else None, alpha_all[e], mean1, var1, kss, work_all[e])
                assert torch.equal(mean1, mean_all[e])
                if want_var:
                    assert torch.equal(var1, var_all[e]), float((var1 - var_all[e]).abs().max())
        # the committee's terms: all experts in one launch == one call per expert, accumulating
        vg = dev(0.2 + rng.random(m), dtype)
        mean_l, var_l = dev(rng.standard_normal((nexp, mpad)), dtype), dev(0.1 + rng.random((nexp, mpad)), dtype)
        for first in (0, -1):
            out_b, out_s = ops.zeros(3, m), ops.zeros(3, m)
            beta_b, prec_b = ops.empty(nexp, m), ops.empty(nexp, m)
            beta_s, prec_s = ops.empty(nexp, m), ops.empty(nexp, m)
            ops.grbcm_local_terms_batched(mean_l, var_l, vg, first, True, out_b, beta_b, prec_b)
            for e in range(nexp):
                ops.grbcm_local_terms(mean_l[e, :m].contiguous(), var_l[e, :m].contiguous(), vg, e == first, True, out_s, beta_s[e], prec_s[e])
            assert torch.equal(out_b, out_s) and torch.equal(beta_b, beta_s) and torch.equal(prec_b, prec_s)
            ref = sum(orc.grbcm_terms(host(mean_l[e, :m]), host(var_l[e, :m]), host(vg), e == first) for e in range(nexp))
            np.testing.assert_allclose(host(out_b), ref, rtol=1e-5 if dtype == torch.float32 else 1e-13)


# --------------------------------------------------------------------------- matrix-pipe bodies (kmfma.hip)
def _grad_inputs(ops, covs, hp, x, y, dtype):
    """K^-1 (lower) and alpha of the model on the device, in `dtype`."""
    from pygpr_amd._ops import pad_to

    n, d = x.shape
    npad = pad_to(n)
    spec = _spec(covs, d)
    hpd, xd = dev(hp), dev(x, dtype)
    k = ops.empty(npad, npad, dtype=dtype)
    invd = ops.potrf_workspace(npad, dtype)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    minv = ops.zeros(npad, npad, dtype=dtype)
    ops.build_factor(spec, hpd, xd, k, invd, info, minv)
    assert int(info.item()) == 0
    ypad = ops.zeros(npad, dtype=dtype)
    ypad[:n] = dev(y, dtype)
    u, alpha = ops.empty(npad, dtype=dtype), ops.empty(npad, dtype=dtype)
    ops.trmv(minv, ypad, u, 0)
    ops.trmv(minv, u, alpha, 1, ops.empty((npad // 256 + 1) * npad, dtype=dtype))
    kinv = ops.zeros(npad, npad, dtype=dtype)
    ops.lauum(minv, kinv)
    return spec, hpd, xd, kinv, alpha


@pytest.mark.parametrize("kind", ["se", "m52"])
@pytest.mark.parametrize("d", [2, 5, 8, 13, 16])
def test_matrix_pipe_bodies_against_the_valu_bodies_and_the_oracle(ops, monkeypatch, kind, d):
    """kmfma.hip (round 5): the covariance build and the gradient contraction with their pairwise products on the matrix pipe, against
    (i) the VALU bodies of kbuild.hip on the same inputs (PG_KB_MFMA=0 / PG_GRAD_MFMA=0) and (ii) the direct-difference oracle -- fp64 at
    the parity tolerances of the rest of the suite, fp32 at the 1e-3 class; mirrored, lower-only and cross builds of one configuration
    agree bit for bit; the padding is the identity / zero."""
    from pygpr_amd._ops import pad_to

    rng = np.random.default_rng(100 * d + len(kind))
    n, m = 333, 200
    x, y = orc.synth(n, d, seed=d)
    xp = rng.random((m, d))
    covs = [orc.SE if kind == "se" else orc.M52, orc.WN]
    hp = np.concatenate([[1.2], 0.4 + 0.8 * rng.random(d), [0.1]])
    spec, npad, mpad = _spec(covs, d), pad_to(n), pad_to(m)
    ref = orc.kernel(covs, hp, x, form="direct") + 1e-7 * np.eye(n)
    ref_x = orc.kernel(covs, hp, x, xp, form="direct")            # [m, n]
    for dtype, tol in ((torch.float64, 2e-14), (torch.float32, 4e-6)):
        hpd, xd, xpd = dev(hp), dev(x, dtype), dev(xp, dtype)
        out = {}
        for mode in ("2", "0"):
            monkeypatch.setenv("PG_KB_MFMA", mode)
            full, low, cross = ops.empty(npad, npad, dtype=dtype), ops.zeros(npad, npad, dtype=dtype), ops.empty(mpad, npad, dtype=dtype)
            ops.kernel_build(spec, hpd, xd, None, full, jitter=1e-7)
            ops.kernel_build(spec, hpd, xd, None, low, lower_only=True, jitter=1e-7)
            ops.kernel_build(spec, hpd, xpd, xd, cross)
            out[mode] = (host(full), host(low), host(cross))
        monkeypatch.delenv("PG_KB_MFMA")
        full, low, cross = out["2"]
        np.testing.assert_allclose(full[:n, :n], ref, atol=tol, rtol=tol)
        np.testing.assert_allclose(cross[:m, :n], ref_x, atol=tol, rtol=tol)
        np.testing.assert_allclose(full, out["0"][0], atol=tol, rtol=tol)
        np.testing.assert_allclose(cross, out["0"][2], atol=tol, rtol=tol)
        assert np.array_equal(full[:n, :n], full[:n, :n].T)                       # exactly symmetric
        pad_ref = np.eye(npad); pad_ref[:n, :n] = full[:n, :n]
        assert np.array_equal(full, pad_ref)                                      # identity padding
        assert not cross[m:, :].any() and not cross[:, n:].any()                  # zero padding of a cross build
        tl = np.tril_indices(npad)
        assert np.array_equal(low[tl], full[tl])                                  # lower-only == mirrored on the lower triangle, bit for bit
        assert not low[:64, 64:].any()                                            # tiles above the diagonal untouched
        dgv = np.float64(np.float32(1.2 ** 2 + 0.1 ** 2 + 1e-7)) if dtype == torch.float32 else 1.2 ** 2 + 0.1 ** 2 + 1e-7
        np.testing.assert_allclose(np.diag(full)[:n], dgv, rtol=2e-7 if dtype == torch.float32 else 1e-15)
    # gradient contraction
    loss_ref, grad_ref = orc.mle_loss_and_grad(covs, hp, x, y, "kinv", form="direct")
    for dtype, rtol in ((torch.float64, 1e-9), (torch.float32, 3e-3)):
        spec, hpd, xd, kinv, alpha = _grad_inputs(ops, covs, hp, x, y, dtype)
        work = ops.empty(ops.nlml_grad_worksize(n, hp.size))
        got = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("PG_GRAD_MFMA", mode)
            g = ops.zeros(hp.size)
            ops.nlml_grad(spec, hpd, xd, n, kinv, alpha, g, work)
            got[mode] = host(g)
        monkeypatch.delenv("PG_GRAD_MFMA")
        scale = np.abs(grad_ref).max()
        np.testing.assert_allclose(got["1"], got["0"], rtol=rtol, atol=rtol * scale)
        np.testing.assert_allclose(got["1"], grad_ref, rtol=10 * rtol if dtype == torch.float64 else rtol * 3, atol=(10 * rtol if dtype == torch.float64 else rtol * 3) * scale)


def test_matrix_pipe_bodies_on_uncentred_and_batched_data(ops, monkeypatch):
    """(i) Points shifted by 1e4: the matrix-pipe bodies work on coordinates relative to the first point, so build and gradient keep their
    accuracy (the VALU fast body's expansion loses eps |x l|^2 there, the direct differences nothing); (ii) batched experts (blockIdx.z /
    blockIdx.y = expert) give each expert the numbers of its one-expert call bit for bit."""
    from pygpr_amd._ops import pad_to

    rng = np.random.default_rng(5)
    n, d = 300, 16
    x, y = orc.synth(n, d, seed=3)
    covs = [orc.SE, orc.WN]
    hp = np.concatenate([[1.1], 0.4 + 0.3 * rng.random(d), [0.1]])
    xs = x + 1.0e4
    npad = pad_to(n)
    k = ops.empty(npad, npad)
    ops.kernel_build(_spec(covs, d), dev(hp), dev(xs), None, k, jitter=1e-7)
    ref = orc.kernel(covs, hp, x, form="direct") + 1e-7 * np.eye(n)
    np.testing.assert_allclose(host(k)[:n, :n], ref, atol=5e-11, rtol=0)          # the shift's own rounding (eps x 1e4 per coordinate), not eps x 1e8
    _, grad_ref = orc.mle_loss_and_grad(covs, hp, x, y, "kinv", form="direct")
    spec, hpd, xd, kinv, alpha = _grad_inputs(ops, covs, hp, x, y, torch.float64)
    work = ops.empty(ops.nlml_grad_worksize(n, hp.size))
    g = ops.zeros(hp.size)
    ops.nlml_grad(spec, hpd, dev(xs), n, kinv, alpha, g, work)
    np.testing.assert_allclose(host(g), grad_ref, rtol=1e-7, atol=1e-7 * np.abs(grad_ref).max())
    # batched: three experts' gradients in one call against three calls
    nexp = 3
    hp_all = np.stack([hp * (1.0 + 0.05 * e) for e in range(nexp)])
    x_all = np.stack([orc.synth(n, d, seed=10 + e)[0] for e in range(nexp)])
    kinv_all, alpha_all = ops.zeros(nexp, npad, npad), ops.zeros(nexp, npad)
    singles = []
    for e in range(nexp):
        ye = orc.synth(n, d, seed=10 + e)[1]
        spec, hpd_e, xd_e, kinv_e, alpha_e = _grad_inputs(ops, covs, hp_all[e], x_all[e], ye, torch.float64)
        kinv_all[e].copy_(kinv_e); alpha_all[e].copy_(alpha_e)
        g = ops.zeros(hp.size)
        ops.nlml_grad(spec, hpd_e, xd_e, n, kinv_e, alpha_e, g, work)
        singles.append(host(g))
    grad_all = ops.zeros(nexp, hp.size)
    xd_all = dev(x_all)
    ops.nlml_grad_batched(spec, dev(hp_all), xd_all, xd_all.stride(0), n, kinv_all, alpha_all, grad_all, ops.empty(nexp * ops.nlml_grad_worksize(n, hp.size)))
    assert np.array_equal(host(grad_all), np.stack(singles))
    # (iii) the contraction's grid: tile column on grid.x (default: K^-1 read in whole row bands) against the first version's mapping, and
    # every walk length -- the same partial sums in the same order, bit for bit
    for grid, gch in (("0", "16"), ("1", "2"), ("1", "5"), ("0", "3"), ("1", "64")):
        monkeypatch.setenv("PG_GRAD_GRID", grid)
        monkeypatch.setenv("PG_GRAD_GCH", gch)
        g2 = ops.zeros(nexp, hp.size)
        ops.nlml_grad_batched(spec, dev(hp_all), xd_all, xd_all.stride(0), n, kinv_all, alpha_all, g2, ops.empty(nexp * ops.nlml_grad_worksize(n, hp.size)))
        np.testing.assert_allclose(host(g2), host(grad_all), rtol=1e-13, atol=0)
    monkeypatch.delenv("PG_GRAD_GRID"); monkeypatch.delenv("PG_GRAD_GCH")


@pytest.mark.parametrize("variant_name", ["GEMM_NT", "GEMM_NT_64"])
@pytest.mark.parametrize("tm,tn", [(5, 3), (20, 3), (23, 9), (9, 8)])
def test_gemm_trapezoid_tiles(ops, variant_name, tm, tn):
    """Round 5: tri = 1 with N < M is a trapezoid -- the lower triangle of the leading N x N block and the full tile rows below it (the
    restricted trailing update and the deferred block's column panels of the factorisation).  Against NumPy on the tiles it covers,
    bit for bit against the same product as a square triangle + a rectangle, and nothing above the diagonal tiles is touched."""
    from pygpr_amd import _lib

    variant = getattr(_lib, variant_name)
    g = torch.Generator(device="cuda").manual_seed(100 * tm + tn)
    m, n, k = 128 * tm, 128 * tn, 256
    a = torch.randn(m, k, device="cuda", dtype=torch.float64, generator=g)
    c0 = torch.randn(m, n, device="cuda", dtype=torch.float64, generator=g)
    c1, c2 = c0.clone(), c0.clone()
    ops.gemm_raw(variant, m, n, k, -1.0, a, a[:n], 1.0, c1, tri=1)
    ops.gemm_raw(variant, n, n, k, -1.0, a[:n], a[:n], 1.0, c2[:n], tri=1)
    ops.gemm_raw(variant, m - n, n, k, -1.0, a[n:], a[:n], 1.0, c2[n:])
    bs = 128 if variant_name == "GEMM_NT" else 64
    mask = torch.ones(m, n, dtype=torch.bool, device="cuda")
    for i in range(0, n, bs):
        mask[i:i + bs, i + bs:] = False                       # tiles strictly above the diagonal
    assert torch.equal(c1[mask], c2[mask])
    assert torch.equal(c1[~mask], c0[~mask])
    ref = host(c0) - host(a) @ host(a[:n]).T
    low = np.tril(np.ones((m, n), dtype=bool))
    np.testing.assert_allclose(host(c1)[low], ref[low], atol=1e-11)


@pytest.mark.parametrize("n", [6144, 8192])
def test_potrf_deferred_trailing_block(ops, n):
    """Round 5, experimental schedule (pg_set_deferred_block; off by default -- measured slower): the coupled chain's first half (columns
    left of about n / 2) updates only the columns up to one panel past the split; the block right of it takes those updates later as
    K = n/2-deep products beside the second half's chain (pg_last_deferred_panels > 0).  The factor agrees with LAPACK's and with the
    default schedule's to rounding; the fused inverse is the factor's inverse."""
    rng = np.random.default_rng(n)
    a = spd(n, rng)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    invd = ops.potrf_workspace(n, torch.float64)
    ad0 = dev(a)
    ops.potrf(ad0, invd, info)
    assert ops.last_deferred_panels() == 0
    ref = np.linalg.cholesky(a)
    scale = np.abs(ref).max()
    ops.set_deferred_block(1)
    try:
        ad = dev(a)
        ops.potrf(ad, invd, info)
        assert int(info.item()) == 0
        if ops.coupled_chain():
            assert ops.last_deferred_panels() > 0
        L = np.tril(host(ad))
        np.testing.assert_allclose(L, ref, rtol=0, atol=1e-11 * scale)
        np.testing.assert_allclose(L, np.tril(host(ad0)), rtol=0, atol=1e-11 * scale)
        ad3, minv = dev(a), ops.empty(n, n)
        ops.potrf_trtri(ad3, invd, info, minv)
        assert int(info.item()) == 0
        mi = np.tril(host(minv))
        r = mi[-512:] @ ref - np.eye(n)[-512:]
        assert np.abs(r).max() < 1e-9
    finally:
        ops.set_deferred_block(0)


@pytest.mark.parametrize("n", [512, 1536, 4096])
def test_factor_ignores_the_upper_triangle(ops, n):
    """ADVICE (round 4): with tri = 1 the updates leave what lies strictly above the diagonal unspecified (whole 128 x 128 diagonal tiles
    update their upper quarter, 64 x 64 tiles and a mixed launch's quarter tiles do not).  No consumer may read it: a matrix whose strictly
    upper triangle is POISONED (NaN) factorises to the same bits as the clean one -- the lower triangle of the factor, the diagonal
    blocks' inverses, the fused triangular inverse and K^-1 = L^-T L^-1."""
    rng = np.random.default_rng(7 + n)
    a = spd(n, rng)
    poisoned = a.copy()
    poisoned[np.triu_indices(n, 1)] = np.nan
    out = []
    for m in (a, poisoned):
        ad, minv, kinv = dev(m), ops.zeros(n, n), ops.zeros(n, n)
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        invd = ops.potrf_workspace(n, torch.float64)
        ops.potrf_trtri(ad, invd, info, minv)
        assert int(info.item()) == 0
        ops.lauum(minv, kinv)
        out.append((np.tril(host(ad)), host(invd[: n * 128]).copy(), np.tril(host(minv)), np.tril(host(kinv))))
    for clean, dirty in zip(*out):
        assert not np.isnan(dirty).any()
        assert np.array_equal(clean, dirty)
    np.testing.assert_allclose(out[1][0], np.linalg.cholesky(a), atol=1e-11)
