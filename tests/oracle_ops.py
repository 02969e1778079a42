"""TEST DOUBLE of pygpr_amd._ops.HipOps for the CPU test tier: the same method surface on CPU tensors,
every number produced by NumPy/SciPy (the oracle's arithmetic).  It exists so that the HOST logic --
shapes, broadcasting, padding, error conventions, optimiser drivers, the multi-rank reductions -- can
be exercised where there is no GPU.  It lives under tests/ and is never imported by the product; tests
install it by monkeypatching `pygpr_amd._ops._OPS`."""
import numpy as np
import scipy.linalg as sla
import torch

from oracle import pygpr_oracle as orc

NB = 128


def _np(t):
    return t.detach().numpy()


def _passes(spec):
    return spec if isinstance(spec, (list, tuple)) else [spec]


def _covs(spec, hp, d):
    """(kind, hp slice) list from a pg_covspec or the list of passes of a long Compose."""
    comps, noise = [], []
    for sp in _passes(spec):
        comps += [({0: "se", 1: "matern52", 2: "sqdist"}[sp.kind[c]], hp[sp.off[c]: sp.off[c] + d + 1]) for c in range(sp.ncomp)]
        noise += [hp[sp.noise_off[i]] for i in range(sp.nnoise)]
    return comps, noise


def _k(comps, xr, xc):
    out = np.zeros((xr.shape[0], xc.shape[0]))
    for kind, h in comps:
        if kind == "se":
            out += orc.se_kernel(h, xc, xr, form="direct")
        elif kind == "sqdist":
            out += (((xr[:, None, :] - xc[None, :, :]) * h[1:]) ** 2).sum(2)
        else:
            out += orc.matern52_kernel(h, xc, xr)
    return out


class OracleOps:
    device = torch.device("cpu")

    def empty(self, *shape, dtype=torch.float64):
        return torch.full(tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape,
                          float("nan"), dtype=dtype)

    def zeros(self, *shape, dtype=torch.float64):
        return torch.zeros(*shape, dtype=dtype)

    def to_device(self, t, dtype=None):
        return t.detach().to(dtype=dtype if dtype is not None else t.dtype).contiguous().clone()

    # covariance
    def kernel_build(self, spec, hp, xr, xc, out, lower_only=False, jitter=0.0):
        h, x = _np(hp), _np(xr).astype(np.float64)
        comps, noise = _covs(spec, h, x.shape[1])
        o = _np(out)
        if xc is None:
            n = x.shape[0]
            full = np.eye(o.shape[0])
            full[:n, :n] = _k(comps, x, x) + (sum(s * s for s in noise) + jitter) * np.eye(n)
            if lower_only:
                mask = np.tril(np.ones_like(full, dtype=bool))
                o[mask] = full[mask]
            else:
                o[...] = full
        else:
            xq = _np(xc).astype(np.float64)
            o[...] = 0.0
            o[: x.shape[0], : xq.shape[0]] = _k(comps, x, xq)
        return out

    def kernel_build_batched(self, spec, hp_all, xr, xc_all, out_all, lower_only=False, jitter=0.0):
        for e in range(out_all.shape[0]):
            xr_e = xr if xr.dim() == 2 else xr[e % xr.shape[0]]
            xc_e = None if xc_all is None else (xc_all if xc_all.dim() == 2 else xc_all[e % xc_all.shape[0]])
            self.kernel_build(spec, hp_all[e % hp_all.shape[0]], xr_e, xc_e, out_all[e], lower_only, jitter)
        return out_all

    def kernel_grad_build(self, spec, hp, x, out):
        h, xx = _np(hp), _np(x).astype(np.float64)
        d = xx.shape[1]
        o = _np(out)
        o[...] = 0.0
        for sp in _passes(spec):
            for c in range(sp.ncomp):
                sl = slice(sp.off[c], sp.off[c] + d + 1)
                fn = orc.se_kernel_and_grad if sp.kind[c] == 0 else orc.matern52_kernel_and_grad
                o[sl] = fn(h[sl], xx, **({"form": "direct"} if sp.kind[c] == 0 else {}))[1]
            for i in range(sp.nnoise):
                o[sp.noise_off[i]] = 2.0 * h[sp.noise_off[i]] * np.eye(xx.shape[0])
        return out

    def sqdist(self, xr, xc, out):
        a = _np(xr).astype(np.float64)
        b = a if xc is None else _np(xc).astype(np.float64)
        o = _np(out)
        o[...] = np.eye(o.shape[0], o.shape[1]) if xc is None else 0.0
        o[: a.shape[0], : b.shape[0]] = ((a[:, None, :] - b[None, :, :]) ** 2).sum(2)
        return out

    # factorisation
    def potrf_workspace(self, n_pad, dtype):
        return torch.zeros(n_pad * NB, dtype=dtype)

    def potrf(self, a, invd, info):
        m = np.tril(_np(a).astype(np.float64))
        m = m + np.tril(m, -1).T
        c, inf = sla.lapack.dpotrf(m, lower=1)
        if not np.isfinite(m).all():
            inf = inf or 1
        info[0] = int(inf)
        if inf == 0:
            low = np.tril(c)
            idx = np.tril_indices(m.shape[0])
            _np(a)[idx] = low[idx]

    def alpha_batched(self, minv_all, y_all, u_all, alpha_all, work_all):
        for e in range(minv_all.shape[0]):
            self.trmv(minv_all[e], y_all[e % y_all.shape[0]], u_all[e], 0)
            self.trmv(minv_all[e], u_all[e], alpha_all[e], 1, work_all[e])

    def potrf_worksize(self, n_pad, dtype):
        return n_pad * NB

    # the experts-together gradient path (MLE on a batched model): per expert through the single-expert doubles
    def alpha_nlml_batched(self, minv_all, y_all, u_all, alpha_all, work_all, n, out_all):
        for e in range(minv_all.shape[0]):
            y = y_all[e % y_all.shape[0]]
            self.trmv(minv_all[e], y, u_all[e], 0)
            self.trmv(minv_all[e], u_all[e], alpha_all[e], 1)
            d = np.diag(_np(minv_all[e]).astype(np.float64))[:n]
            out_all[e, 0] = 0.5 * float(_np(y)[:n].astype(np.float64) @ _np(alpha_all[e])[:n].astype(np.float64)) \
                - float(np.log(d).sum()) + 0.5 * n * np.log(2 * np.pi)

    def lauum_batched(self, minv_all, kinv_all):
        for e in range(minv_all.shape[0]):
            self.lauum(minv_all[e], kinv_all[e])

    def nlml_grad_batched(self, spec, hp_all, x_all, x_stride, n, kinv_all, alpha_all, grad_all, work):
        for e in range(kinv_all.shape[0]):
            g = torch.zeros(hp_all.shape[-1], dtype=torch.float64)
            self.nlml_grad(spec, hp_all[e], x_all[e if x_stride else 0], n, kinv_all[e], alpha_all[e], g, None)
            grad_all[e, : g.numel()] = g

    def build_factor_batched(self, spec, hp_all, x_all, x_stride, a_all, invd_all, info_all, minv_all=None, jitter=1e-7):
        for e in range(a_all.shape[0]):
            self.build_factor(spec, hp_all[e], x_all[e if x_stride else 0], a_all[e], invd_all[e], info_all[e: e + 1],
                              minv_all[e] if minv_all is not None else None, jitter)

    def build_factor(self, spec, hp, x, a, invd, info, minv=None, jitter=1e-7):
        self.kernel_build(spec, hp, x, None, a, lower_only=True, jitter=jitter)
        return self.potrf_trtri(a, invd, info, minv) if minv is not None else self.potrf(a, invd, info)

    def potrf_trtri(self, a, invd, info, minv):
        self.potrf(a, invd, info)
        if int(info[0]) == 0:
            self.trtri(a, invd, minv)

    def potrs_vec(self, chol, invd, y, x, work=None):
        x.copy_(torch.from_numpy(sla.cho_solve((np.tril(_np(chol).astype(np.float64)), True), _np(y).astype(np.float64),
                                              check_finite=False)))

    def trtri(self, chol, invd, minv):
        low = np.tril(_np(chol).astype(np.float64))
        minv.copy_(torch.from_numpy(sla.solve_triangular(low, np.eye(low.shape[0]), lower=True, check_finite=False)))

    def lauum(self, minv, kinv):
        m = np.tril(_np(minv).astype(np.float64))
        kinv.copy_(torch.from_numpy(np.tril(m.T @ m)))

    def trmv(self, minv, x, y, trans, work=None):
        m = np.tril(_np(minv).astype(np.float64))
        y.copy_(torch.from_numpy((m.T if trans else m) @ _np(x).astype(np.float64)))

    def tril(self, a, n):
        a.copy_(torch.tril(a))

    # NLML
    def nlml_value(self, chol, y, alpha, n, out):
        d = np.diag(_np(chol).astype(np.float64))[:n]
        out[0] = 0.5 * float(_np(y)[:n].astype(np.float64) @ _np(alpha)[:n].astype(np.float64)) \
            + float(np.log(d).sum()) + 0.5 * n * np.log(2 * np.pi)

    def alpha_nlml_async(self, chol, minv, y, u, alpha, work, n, out):
        self.trmv(minv, y, u, 0)
        self.trmv(minv, u, alpha, 1)
        self.nlml_value(chol, y, alpha, n, out)

    def nlml_grad_worksize(self, n, nhp):
        return 1

    def nlml_grad(self, spec, hp, x, n, kinv, alpha, grad, work):
        h, xx = _np(hp), _np(x).astype(np.float64)
        d = xx.shape[1]
        ki = np.tril(_np(kinv).astype(np.float64))[:n, :n]
        ki = ki + np.tril(ki, -1).T
        al = _np(alpha).astype(np.float64)[:n]
        w = ki - np.outer(al, al)
        dk = np.zeros((grad.numel(), n, n))
        self.kernel_grad_build(spec, hp, x, torch.from_numpy(dk))
        grad.copy_(torch.from_numpy(0.5 * np.einsum("ij,kij->k", w, dk)))

    # prediction
    def predict_mean_q(self, ks, minv, alpha, mean, var, kss, work):
        k = _np(ks).astype(np.float64)
        mean.copy_(torch.from_numpy(k.T @ _np(alpha).astype(np.float64)))
        if var is not None:
            v = np.tril(_np(minv).astype(np.float64)) @ k
            var.copy_(torch.from_numpy(kss - (v * v).sum(0)))

    def predict_mean_q_kt(self, kt, minv, alpha, mean, var, kss, work):
        self.predict_mean_q(torch.from_numpy(np.ascontiguousarray(_np(kt).T)), minv, alpha, mean, var, kss, work)

    def predict_mean_q_kt_batched(self, kt_all, minv_all, alpha_all, mean_all, var_all, spec, hp_all, work_all):
        for e in range(kt_all.shape[0]):
            h = _np(hp_all[e % hp_all.shape[0]])
            kss = 0.0
            for sp in _passes(spec):
                kss += sum(h[sp.off[c]] ** 2 for c in range(sp.ncomp)) + sum(h[sp.noise_off[i]] ** 2 for i in range(sp.nnoise))
            self.predict_mean_q_kt(kt_all[e], minv_all[e] if var_all is not None else None, alpha_all[e], mean_all[e],
                                   var_all[e] if var_all is not None else None, kss, None)

    def trmm_lower(self, minv, ks, v):
        v.copy_(torch.from_numpy(np.tril(_np(minv).astype(np.float64)) @ _np(ks).astype(np.float64)))

    def syrk_tn_sub(self, v, c, lower_only=True):
        vv = _np(v).astype(np.float64)
        c -= torch.from_numpy(vv.T @ vv)

    def trmm_lower_kt(self, minv, kt, vt):
        if kt.dim() == 3:
            for e in range(kt.shape[0]):
                self.trmm_lower_kt(minv[e], kt[e], vt[e])
            return
        vt.copy_(torch.from_numpy(_np(kt).astype(np.float64) @ np.tril(_np(minv).astype(np.float64)).T))

    def syrk_nt_sub_batched(self, vt_all, c_all, lower_only=True):
        for e in range(vt_all.shape[0]):
            vv = _np(vt_all[e]).astype(np.float64)
            c_all[e] -= torch.from_numpy(vv @ vv.T)

    # grBCM
    def grbcm_local_terms(self, mean_c, var_c, var_g, is_first, accumulate, out, beta=None, prec=None):
        t = orc.grbcm_terms(_np(mean_c).astype(np.float64), _np(var_c).astype(np.float64), _np(var_g).astype(np.float64), is_first)
        if accumulate:
            out += torch.from_numpy(t)
        else:
            out.copy_(torch.from_numpy(t))
        if beta is not None:
            beta.copy_(torch.from_numpy(t[0]))
        if prec is not None:
            prec.copy_(torch.from_numpy(1.0 / _np(var_c).astype(np.float64)))

    def grbcm_local_terms_batched(self, mean_all, var_all, var_g, first, accumulate, out, beta=None, prec=None):
        m = var_g.numel()
        for c in range(mean_all.shape[0]):
            self.grbcm_local_terms(mean_all[c, :m], var_all[c, :m], var_g, c == first, accumulate or c > 0, out,
                                   beta[c] if beta is not None else None, prec[c] if prec is not None else None)

    def grbcm_finish(self, sums, mean_g, var_g, mean, var, beta0=None, prec0=None):
        mu, v = orc.grbcm_finish(_np(sums), _np(mean_g).astype(np.float64), _np(var_g).astype(np.float64))
        mean.copy_(torch.from_numpy(mu))
        var.copy_(torch.from_numpy(v))
        if beta0 is not None:
            beta0.copy_(1.0 - sums[0])
        if prec0 is not None:
            prec0.copy_(torch.from_numpy(1.0 / _np(var_g).astype(np.float64)))

    # full-covariance committee
    def grbcm_weighted_prec(self, prec, beta, acc, m, accumulate):
        p, b, a = _np(prec).astype(np.float64), _np(beta), _np(acc)
        w = np.tril(0.5 * (b[:, None] + b[None, :]) * p[:m, :m])
        if not accumulate:
            a[...] = np.eye(a.shape[0])
            a[:m, :m] = 0.0
        a[:m, :m] += w

    def symmetrize(self, a, n):
        low = np.tril(_np(a)[:n, :n])
        _np(a)[:n, :n] = low + np.tril(low, -1).T

    def grbcm_finish_full(self, sums, mean_g, var_g, cov, mean):
        s = _np(sums)
        pg_ = 1.0 / _np(var_g).astype(np.float64)
        mean.copy_(torch.from_numpy(np.diag(_np(cov))[: s.shape[1]] * (s[2] + (1.0 - s[0]) * pg_ * _np(mean_g).astype(np.float64))))

    def spd_inverse_lower(self, a_pad):
        info = torch.zeros(1, dtype=torch.int32)
        self.potrf(a_pad, None, info)
        minv = torch.zeros_like(a_pad)
        self.trtri(a_pad, None, minv)
        out = torch.zeros_like(a_pad)
        self.lauum(minv, out)
        return out, info

    def spd_inverse_lower_batched(self, a_all):
        info = torch.zeros(a_all.shape[0], dtype=torch.int32)
        for e in range(a_all.shape[0]):
            out, i = self.spd_inverse_lower(a_all[e].clone())
            a_all[e].copy_(out)
            info[e] = i[0]
        return a_all, info

    def sqdist_argmin(self, x, centres, dist=None, idx=None):
        xx, cc = _np(x).astype(np.float64), _np(centres).astype(np.float64)
        d2 = ((xx[:, None, :] - cc[None, :, :]) ** 2).sum(2)
        if dist is not None:
            dist.copy_(torch.from_numpy(d2))
        if idx is not None:
            idx.copy_(torch.from_numpy(np.argmin(d2, axis=1).astype(np.int32)))
