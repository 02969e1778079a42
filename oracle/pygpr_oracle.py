"""
CPU ORACLE for the PyGPR dense-GP hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This file is a NumPy/SciPy restatement of the reference's algorithm
(sarath-srinivas/PyGPR, all citations relative to /root/reference/).  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it, and there only as the checker / the CPU baseline.  Nothing under
`pygpr_amd/` imports it; the product path raises when the HIP library is
missing instead of falling back to this file.

Parity pin: every function below is checked in `tests/test_oracle_golden.py`
against `tests/golden/*.npz`, which were produced by importing the reference
itself in the authoring container (`tests/golden/make_golden.py`).  The one
exception is `matern52_*`: the reference has no Matern covariance kernel
(SURVEY.md section 8 row a-13), so that part is "parity unpinned by the
reference" and is pinned by sklearn's Matern(nu=2.5) and finite differences
instead.

Two flavours are kept for the O(n^3) gradient:
  * `mle_loss_and_grad(..., route="solve")`  -- reference-faithful: materialise
    dK[nhp,n,n] and cho_solve it (PyGPR/loss.py:92-128), cost n^3/3 + 2 nhp n^3;
  * `mle_loss_and_grad_as_written` -- the same as-written algorithm on torch CPU (second bench baseline),
  * `route="kinv"` -- the lean K^-1 route (g_k = 1/2 sum (K^-1 - a a^T) o dK_k),
    the same mathematics the HIP path runs; cost ~ n^3.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla

JITTER = 1e-7  # PyGPR/gpr.py:68, PyGPR/loss.py:38,63,96


# --------------------------------------------------------------------------
# covariance kernels (PyGPR/covar.py)
# --------------------------------------------------------------------------
class SE:
    """Squared exponential, hp = [sig, l_1..l_d], l = INVERSE length scales.
    PyGPR/covar.py:84-206."""

    kind = "se"

    @staticmethod
    def nhp(d):  # covar.py:89-94
        return d + 1

    @staticmethod
    def init(d):  # covar.py:96-100
        return np.ones(d + 1)


class M52:
    """Matern-5/2 with the SE hp layout.  Not in the reference."""

    kind = "matern52"

    @staticmethod
    def nhp(d):
        return d + 1

    @staticmethod
    def init(d):
        return np.ones(d + 1)


class WN:
    """White noise sig_n^2 I, hp = [sig_n].  PyGPR/covar.py:209-269."""

    kind = "wn"

    @staticmethod
    def nhp(d):  # covar.py:214-219
        return 1

    @staticmethod
    def init(d):  # covar.py:221-225
        return 1e-4 * np.ones(1)


def _sqdist(xl, xpl=None, form="gemm"):
    """Squared distances of pre-scaled inputs.  form="gemm" follows
    PyGPR/covar.py:102-127 (-2 X X^T + |x|^2 + |x'|^2); form="direct" sums
    squared differences (what the HIP kernel does)."""
    if form == "gemm":
        x2 = np.sum(xl * xl, axis=1)
        if xpl is None:
            return -2.0 * (xl @ xl.T) + (x2[:, None] + x2[None, :])
        xp2 = np.sum(xpl * xpl, axis=1)
        return -2.0 * (xpl @ xl.T) + (xp2[:, None] + x2[None, :])
    a = xl if xpl is None else xpl
    diff = a[:, None, :] - xl[None, :, :]
    return np.sum(diff * diff, axis=2)


def se_kernel(hp, x, xp=None, form="gemm"):
    """PyGPR/covar.py:129-167.  Returns [n,n] or [m,n] (rows = test)."""
    assert hp.shape[-1] == x.shape[-1] + 1
    sig, ls = hp[0], hp[1:]
    xl = x * ls
    xpl = None if xp is None else xp * ls
    sqd = _sqdist(xl, xpl, form)
    return sig * sig * np.exp(-sqd)


def se_kernel_and_grad(hp, x, form="gemm"):
    """PyGPR/covar.py:169-206: dK/dsig = 2K/sig; dK/dl_k = -2 l_k (x_k-x'_k)^2 K."""
    k = se_kernel(hp, x, form=form)
    sig, ls = hp[0], hp[1:]
    n, d = x.shape
    dk = np.empty((d + 1, n, n))
    dk[0] = k * (2.0 / sig)
    for a in range(d):
        diff = x[:, a][:, None] - x[:, a][None, :]
        dk[a + 1] = -2.0 * ls[a] * diff * diff * k
    return k, dk


def matern52_kernel(hp, x, xp=None):
    """K = sig^2 (1 + sqrt5 r + 5 r^2/3) exp(-sqrt5 r), r = |(x-x') o l|.
    SURVEY.md 8 a-13 (new work, no reference counterpart)."""
    assert hp.shape[-1] == x.shape[-1] + 1
    sig, ls = hp[0], hp[1:]
    r2 = np.maximum(_sqdist(x * ls, None if xp is None else xp * ls, "direct"), 0.0)
    r = np.sqrt(r2)
    s5 = np.sqrt(5.0)
    return sig * sig * (1.0 + s5 * r + (5.0 / 3.0) * r2) * np.exp(-s5 * r)


def matern52_kernel_and_grad(hp, x):
    """dK/dl_k = -(5/3) sig^2 (1 + sqrt5 r) exp(-sqrt5 r) l_k D_k^2."""
    sig, ls = hp[0], hp[1:]
    n, d = x.shape
    r2 = np.maximum(_sqdist(x * ls, None, "direct"), 0.0)
    r = np.sqrt(r2)
    s5 = np.sqrt(5.0)
    e = np.exp(-s5 * r)
    k = sig * sig * (1.0 + s5 * r + (5.0 / 3.0) * r2) * e
    dk = np.empty((d + 1, n, n))
    dk[0] = k * (2.0 / sig)
    base = -(5.0 / 3.0) * sig * sig * (1.0 + s5 * r) * e
    for a in range(d):
        diff = x[:, a][:, None] - x[:, a][None, :]
        dk[a + 1] = base * ls[a] * diff * diff
    return k, dk


def wn_kernel(hp, x, xp=None):
    """PyGPR/covar.py:227-245; with xp given the reference returns int tensor(0)."""
    if xp is not None:
        return np.int64(0)
    return hp[0] * hp[0] * np.eye(x.shape[0])


def wn_kernel_and_grad(hp, x):
    """PyGPR/covar.py:247-269: dK/dsig_n = 2 sig_n I."""
    n = x.shape[0]
    return wn_kernel(hp, x), (2.0 * hp[0] * np.eye(n))[None]


_K = {"se": se_kernel, "wn": wn_kernel, "matern52": matern52_kernel}
_KG = {"se": se_kernel_and_grad, "wn": wn_kernel_and_grad,
       "matern52": matern52_kernel_and_grad}


def _chunks(covs, d):
    sizes = [c.nhp(d) for c in covs]
    off = np.concatenate([[0], np.cumsum(sizes)])
    return [(int(off[i]), int(off[i + 1])) for i in range(len(covs))]


def nhp(covs, d):
    return sum(c.nhp(d) for c in covs)


def init_params(covs, d):
    """Compose.init_params, PyGPR/covar.py:45-48."""
    return np.concatenate([c.init(d) for c in covs])


def kernel(covs, hp, x, xp=None, form="gemm"):
    """Compose.kernel, PyGPR/covar.py:50-62 (sum of children; hp concatenated in
    list order).  Unbatched: hp[nhp], x[n,d], xp[m,d]."""
    d = x.shape[-1]
    assert hp.shape[-1] == nhp(covs, d)  # covar.py:52
    out = None
    for c, (a, b) in zip(covs, _chunks(covs, d)):
        kw = {"form": form} if c.kind == "se" else {}
        k = _K[c.kind](hp[a:b], x, xp, **kw)
        out = k if out is None else out + k
    return out


def kernel_and_grad(covs, hp, x, form="gemm"):
    """Compose.kernel_and_grad, PyGPR/covar.py:64-81 (dK stacked on dim -3)."""
    d = x.shape[-1]
    assert hp.shape[-1] == nhp(covs, d)  # covar.py:66
    ks, dks = None, []
    for c, (a, b) in zip(covs, _chunks(covs, d)):
        kw = {"form": form} if c.kind == "se" else {}
        k, dk = _KG[c.kind](hp[a:b], x, **kw)
        ks = k if ks is None else ks + k
        dks.append(dk)
    return ks, np.concatenate(dks, axis=0)


# --------------------------------------------------------------------------
# Exact_GP (PyGPR/gpr.py)
# --------------------------------------------------------------------------
def gp_update(covs, hp, x, y, form="gemm"):
    """Exact_GP.update, PyGPR/gpr.py:65-74: K + 1e-7 I, lower Cholesky, alpha."""
    k = kernel(covs, hp, x, form=form)
    k[np.diag_indices_from(k)] += JITTER
    try:
        chol = sla.cholesky(k, lower=True)
    except sla.LinAlgError as e:  # torch raises torch.linalg.LinAlgError
        raise np.linalg.LinAlgError(str(e))
    alpha = sla.cho_solve((chol, True), y)
    return k, chol, alpha


def gp_predict(covs, hp, x, y, xp, var="diag", form="gemm"):
    """Exact_GP.predict / predict_var / predict_covar, PyGPR/gpr.py:76-120.
    K** keeps sig_n^2 on its diagonal (White_noise sees xp=None, gpr.py:98)."""
    _, chol, alpha = gp_update(covs, hp, x, y, form)
    ks = kernel(covs, hp, x, xp, form=form)  # [m,n]
    mean = ks @ alpha
    if var == "diag":
        kss = kernel(covs, hp, xp, form=form)
        lks = sla.cho_solve((chol, True), ks.T)
        return mean, np.diag(kss) - np.sum(ks * lks.T, axis=1)
    if var == "full":
        kss = kernel(covs, hp, xp, form=form)
        lks = sla.cho_solve((chol, True), ks.T)
        return mean, kss - ks @ lks
    return mean, NotImplemented


# --------------------------------------------------------------------------
# MLE (PyGPR/loss.py)
# --------------------------------------------------------------------------
def mle_loss(covs, hp, x, y, form="gemm"):
    """MLE.loss, PyGPR/loss.py:35-57:
    1/2 y^T a + sum log L_ii + n/2 log 2pi."""
    _, chol, alpha = gp_update(covs, hp, x, y, form)
    n = y.shape[-1]
    return 0.5 * float(alpha @ y) + float(np.sum(np.log(np.diag(chol)))) \
        + 0.5 * n * np.log(2.0 * np.pi)


def mle_loss_and_grad(covs, hp, x, y, route="solve", form="gemm"):
    """MLE.loss_and_grad, PyGPR/loss.py:92-128.
    jac_k = -1/2 (a^T dK_k a - tr(K^-1 dK_k))   (loss.py:116-121)."""
    k, dk = kernel_and_grad(covs, hp, x, form=form)
    k[np.diag_indices_from(k)] += JITTER
    chol = sla.cholesky(k, lower=True)
    alpha = sla.cho_solve((chol, True), y)
    n = y.shape[-1]
    loss = 0.5 * float(alpha @ y) + float(np.sum(np.log(np.diag(chol)))) \
        + 0.5 * n * np.log(2.0 * np.pi)
    if route == "solve":  # as written in the reference
        tr1 = np.einsum("i,kij,j->k", alpha, dk, alpha)
        tr2 = np.array([np.trace(sla.cho_solve((chol, True), dk[a]))
                        for a in range(dk.shape[0])])
        return loss, -0.5 * (tr1 - tr2)
    kinv = sla.cho_solve((chol, True), np.eye(n))
    w = kinv - np.outer(alpha, alpha)
    return loss, 0.5 * np.einsum("ij,kij->k", w, dk)


def mle_grad(covs, hp, x, y, route="solve", form="gemm"):
    """MLE.grad, PyGPR/loss.py:59-90 (same maths as loss_and_grad)."""
    return mle_loss_and_grad(covs, hp, x, y, route, form)[1]


def mle_loss_and_grad_lean(hp, x, y):
    """Lean CPU baseline for bench.py (Compose([SE, WN]) only): the K^-1 route on the reference's own
    substrate -- torch CPU fp64, LAPACK potrf/potri through torch.linalg, all intra-op threads -- with NO
    [nhp,n,n] stack.  The per-dimension contraction sum_ij W_ij K_ij (x_ia - x_ja)^2 is expanded to
    2 sum_i x_ia^2 r_i - 2 x_a^T (W o K) x_a (r = row sums), i.e. one n x n x d GEMM instead of d n^2 passes.
    Same numbers as mle_loss_and_grad(route="kinv") to rounding; memory ~ 4 n^2 doubles."""
    import torch

    n, d = x.shape
    xt, yt = torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(np.ascontiguousarray(y))
    sig, sn = float(hp[0]), float(hp[d + 1])
    ls = torch.from_numpy(np.ascontiguousarray(hp[1:d + 1]))
    xl = xt * ls
    x2 = (xl * xl).sum(1)
    kse = torch.addmm(x2[:, None] + x2[None, :], xl, xl.T, alpha=-2.0)      # covar.py:102-127
    kse.neg_().exp_().mul_(sig * sig)                                          # covar.py:147-149
    a = kse.clone()
    a.diagonal().add_(sn * sn + JITTER)
    chol = torch.linalg.cholesky(a)
    del a
    alpha = torch.cholesky_solve(yt[:, None], chol)[:, 0]
    loss = 0.5 * float(alpha @ yt) + float(torch.log(chol.diagonal()).sum()) + 0.5 * n * np.log(2.0 * np.pi)
    w = torch.cholesky_inverse(chol)
    del chol
    w.addr_(alpha, alpha, alpha=-1.0)                                          # W = K^-1 - a a^T
    tr_w = float(w.diagonal().sum())
    w.mul_(kse)                                                                # W o K_se
    del kse
    g = np.empty(d + 2)
    r = w.sum(1)
    g[0] = 0.5 * (2.0 / sig) * float(r.sum())
    wx = w @ xt                                                                # [n, d]
    quad = 2.0 * ((xt * xt) * r[:, None]).sum(0) - 2.0 * (xt * wx).sum(0)      # sum_ij (W o K)_ij (x_ia - x_ja)^2
    g[1:d + 1] = (0.5 * -2.0 * ls * quad).numpy()
    g[d + 1] = 0.5 * 2.0 * sn * tr_w
    return loss, g


def matern52_nlml_lean(hp, x, y, rows=1024):
    """NLML of Compose([Matern52, WN]) (MLE.loss, PyGPR/loss.py:35-57) at sizes where matern52_kernel's [n, n, d]
    difference array does not fit (n = 33792: 146 GB): the same DIRECT squared differences, `rows` rows at a time (torch.cdist without the
    matmul expansion, all intra-op threads), the same covariance formula, LAPACK potrf + a two-sided triangular solve.  Equal to
    mle_loss([M52, WN], ...) to rounding (tests/test_oracle_golden.py); memory ~ 2 n^2 doubles."""
    import torch

    n, d = x.shape
    sig, sn = float(hp[0]), float(hp[d + 1])
    xl = torch.from_numpy(np.ascontiguousarray(x * hp[1:d + 1]))
    yt = torch.from_numpy(np.ascontiguousarray(y))
    k = torch.empty(n, n, dtype=torch.float64)
    s5 = float(np.sqrt(5.0))
    for r0 in range(0, n, rows):
        r = torch.cdist(xl[r0:r0 + rows], xl, p=2.0, compute_mode="donot_use_mm_for_euclid_dist")
        e = torch.exp(-s5 * r)
        k[r0:r0 + rows] = (sig * sig) * (1.0 + s5 * r + (5.0 / 3.0) * r * r) * e
    k.diagonal().add_(sn * sn + JITTER)
    chol = torch.linalg.cholesky(k)
    del k
    alpha = torch.cholesky_solve(yt[:, None], chol)[:, 0]
    return 0.5 * float(alpha @ yt) + float(torch.log(chol.diagonal()).sum()) + 0.5 * n * np.log(2.0 * np.pi)


def mle_loss_and_grad_as_written(hp, x, y):
    """Second CPU baseline for bench.py (Compose([SE, WN]) only): the reference's algorithm AS WRITTEN, on its
    own substrate (torch CPU fp64): the dK stack [nhp,n,n] is materialised (covar.py:169-206, 247-269), and
    tr(K^-1 dK_k) comes from one batched cholesky_solve on the whole stack (loss.py:92-128, solve at :116).
    Cost n^3/3 + 2 nhp n^3.  Same numbers as mle_loss_and_grad(route="solve") to rounding."""
    import torch

    n, d = x.shape
    xt, yt = torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(np.ascontiguousarray(y))
    sig, sn = float(hp[0]), float(hp[d + 1])
    ls = torch.from_numpy(np.ascontiguousarray(hp[1:d + 1]))
    xl = xt * ls
    x2 = (xl * xl).sum(1)
    k = torch.addmm(x2[:, None] + x2[None, :], xl, xl.T, alpha=-2.0)        # covar.py:102-127
    k.neg_().exp_().mul_(sig * sig)                                            # covar.py:147-149
    dk = torch.empty(d + 2, n, n, dtype=torch.float64)
    dk[0] = k * (2.0 / sig)                                                    # covar.py:189
    xc = xt.T.contiguous()
    for a in range(d):                                                         # covar.py:191-199
        diff = xc[a][:, None] - xc[a][None, :]
        dk[a + 1] = diff.mul_(diff).mul_(k).mul_(-2.0 * float(ls[a]))
    dk[d + 1] = torch.eye(n, dtype=torch.float64) * (2.0 * sn)                 # covar.py:262-264
    k.diagonal().add_(sn * sn + JITTER)
    chol = torch.linalg.cholesky(k)
    alpha = torch.cholesky_solve(yt[:, None], chol)[:, 0]
    loss = 0.5 * float(alpha @ yt) + float(torch.log(chol.diagonal()).sum()) + 0.5 * n * np.log(2.0 * np.pi)
    tr1 = torch.einsum("i,kij,j->k", alpha, dk, alpha)
    tr2 = torch.cholesky_solve(dk, chol).diagonal(dim1=-2, dim2=-1).sum(-1)   # loss.py:116-119
    return loss, (-0.5 * (tr1 - tr2)).numpy()


def get_learn_rate(covs, hp, x, y, eps, form="gemm"):
    """hp_update.get_learn_rate, PyGPR/hp_update.py:6-28."""
    f0, jac = mle_loss_and_grad(covs, hp, x, y, "solve", form)
    fp = mle_loss(covs, hp - eps * jac, x, y, form)
    fm = mle_loss(covs, hp + eps * jac, x, y, form)
    c1 = (fp - fm) / (2.0 * eps)
    c2 = (fp + fm - 2 * f0) / (2.0 * eps ** 2)
    return -0.5 * (c1 / c2)


# --------------------------------------------------------------------------
# grBCM (PyGPR/gr_bcm.py)
# --------------------------------------------------------------------------
def grbcm_terms(mean_c, var_c, var_g, is_first):
    """Per-expert terms of GRBCM.aggregate (PyGPR/gr_bcm.py:125-144) that are
    summed across experts: (beta_c, beta_c prec_c, beta_c prec_c mu_c) with
    beta_c = 1/2 (log prec_c - log prec_0), and beta := 1 for the first local
    expert (gr_bcm.py:132)."""
    prec_c = 1.0 / var_c
    prec_0 = 1.0 / var_g
    beta = np.ones_like(var_c) if is_first else 0.5 * (np.log(prec_c) - np.log(prec_0))
    return np.stack([beta, beta * prec_c, beta * prec_c * mean_c])


def grbcm_finish(sums, mean_g, var_g):
    """Finish GRBCM.aggregate from the summed local terms:
    beta_0 = 1 - sum beta_c (gr_bcm.py:133); var = 1/sum(beta prec) (:143);
    mu = var * sum(beta prec mu) (:144)."""
    prec_0 = 1.0 / var_g
    beta_0 = 1.0 - sums[0]
    prec = sums[1] + beta_0 * prec_0
    var = 1.0 / prec
    return var * (sums[2] + beta_0 * prec_0 * mean_g), var


def grbcm_aggregate(mean_g, var_g, mean_l, var_l):
    """GRBCM.aggregate(var="diag"), PyGPR/gr_bcm.py:116-149.  Returns
    (mu[m], var[m], beta[nc+1,m], prec[nc+1,m])."""
    nc = mean_l.shape[0]
    m = mean_g.shape[-1]
    beta = np.empty((nc + 1, m))
    prec = np.empty((nc + 1, m))
    prec[0] = 1.0 / var_g
    prec[1:] = 1.0 / var_l
    beta[1:] = 0.5 * (np.log(prec[1:]) - np.log(prec[0]))
    beta[1] = 1.0
    beta[0] = -(beta[1:].sum(0) - 1.0)
    ys = np.concatenate([mean_g[None], mean_l])
    precs = prec * beta
    var = 1.0 / precs.sum(0)
    mu = (ys * precs).sum(0) * var
    return mu, var, beta, prec


def grbcm_aggregate_full(mean_g, cov_g, mean_l, cov_l):
    """GRBCM.aggregate(var="full") + aggregate_full_covar,
    PyGPR/gr_bcm.py:99-114,116-149."""
    nc = mean_l.shape[0]
    var_g = np.diag(cov_g)
    var_l = np.stack([np.diag(c) for c in cov_l])
    _, _, beta, prec = grbcm_aggregate(mean_g, var_g, mean_l, var_l)
    cov_gl = np.concatenate([cov_g[None], cov_l])
    m = cov_g.shape[-1]
    prec_gl = np.stack([sla.cho_solve((sla.cholesky(c, lower=True), True), np.eye(m))
                        for c in cov_gl])
    bc = 0.5 * (beta[:, :, None] + beta[:, None, :])
    p = (prec_gl * bc).sum(0)
    cov = sla.cho_solve((sla.cholesky(p, lower=True), True), np.eye(m))
    ys = np.concatenate([mean_g[None], mean_l])
    mu = (ys * (prec * beta)).sum(0) * np.diag(cov)
    return mu, cov


def grbcm_data(xl, yl, xg, yg):
    """GRBCM.__init__, PyGPR/gr_bcm.py:12-34: every local expert sees the global
    (communication) set followed by its own shard."""
    nc = xl.shape[0]
    x = np.concatenate([np.broadcast_to(xg, (nc,) + xg.shape), xl], axis=1)
    y = np.concatenate([np.broadcast_to(yg, (nc,) + yg.shape), yl], axis=1)
    return x, y


def grbcm_predict(covs, hp_g, hp_l, xl, yl, xg, yg, xs, var="diag", form="gemm"):
    """GRBCM.predict, PyGPR/gr_bcm.py:151-155.  hp_l is [nc,nhp]."""
    x, y = grbcm_data(xl, yl, xg, yg)
    mg, vg = gp_predict(covs, hp_g, xg, yg, xs, var, form)
    res = [gp_predict(covs, hp_l[c], x[c], y[c], xs, var, form) for c in range(x.shape[0])]
    ml = np.stack([r[0] for r in res])
    vl = np.stack([r[1] for r in res])
    if var == "diag":
        return grbcm_aggregate(mg, vg, ml, vl)
    return grbcm_aggregate_full(mg, vg, ml, vl)


# --------------------------------------------------------------------------
# synthetic inputs shared by tests, smoke() and bench.py (SURVEY.md 8 d)
# --------------------------------------------------------------------------
def synth(n, d, seed=1234, noise=0.1):
    """x ~ U[0,1]^d, y = sin(-sum x) + noise N(0,1)."""
    rng = np.random.default_rng(seed)
    x = rng.random((n, d))
    y = np.sin(-x.sum(1)) + noise * rng.standard_normal(n)
    return x, y


def se_distance(x, xp=None, form="gemm"):
    """Squared_exponential.distance, PyGPR/covar.py:102-127: squared Euclidean distances of the inputs as given
    ([n,n], or [m,n] with rows = xp)."""
    return _sqdist(x, xp, form)
