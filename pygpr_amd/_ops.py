"""Device-op layer: torch-ROCm tensors are only buffers + the current stream; every number is
produced by libpygpr_hip through the C ABI (include/pygpr_hip.h).  No CPU path exists here:
`get_ops()` raises when the library or a GPU is missing."""
import atexit
import ctypes as C
import os

import torch

from . import _lib
from ._lib import PAD, PG_F32, PG_F64, CovSpec

JITTER = 1e-7  # PyGPR/gpr.py:68, loss.py:38,63,96


def pad_to(n, q=PAD):
    return ((int(n) + q - 1) // q) * q


def _code(dtype):
    if dtype == torch.float64:
        return PG_F64
    if dtype == torch.float32:
        return PG_F32
    raise TypeError("pygpr_amd computes in float64 or float32, got %s" % dtype)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class HipOps:
    """One per process (one process per GPU)."""

    def __init__(self):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("pygpr_amd needs a HIP device (MI355X); there is no CPU fallback")
        h = C.c_void_p()
        _lib.check(self.lib.pg_create(C.byref(h)), "pg_create")
        self.h = h
        self.device = torch.device("cuda", torch.cuda.current_device())
        # the library destroys live handles itself at process exit (capi.hip: C atexit registered by pg_create); closing
        # here as well releases the streams while torch's allocator is still up.  PG_NO_PY_ATEXIT=1 leaves it to the library
        # (used once to verify the library-side teardown under rocprofv3).
        if not os.environ.get("PG_NO_PY_ATEXIT"):
            atexit.register(self.close)

    def close(self):
        if getattr(self, "h", None) is not None:
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            self.lib.pg_destroy(self.h)
            self.h = None

    # -- helpers ------------------------------------------------------------------------------
    def _st(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _chk(self, *ts):
        for t in ts:
            if t is None:
                continue
            if not t.is_cuda or not t.is_contiguous():
                raise ValueError("device op needs contiguous CUDA tensors")

    def empty(self, *shape, dtype=torch.float64):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def zeros(self, *shape, dtype=torch.float64):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def to_device(self, t, dtype=None):
        return t.detach().to(device=self.device, dtype=dtype if dtype is not None else t.dtype).contiguous()

    # -- covariance assembly ------------------------------------------------------------------
    def kernel_build(self, spec, hp, xr, xc, out, lower_only=False, jitter=0.0):
        """out[rows_pad, cols_pad] <- k(xr, xc) (xc None: symmetric + noise/jitter diagonal).  `spec` is one pg_covspec
        or the list of passes of a long Compose (make_specs): later passes accumulate."""
        self._chk(hp, xr, xc, out)
        assert hp.dtype == torch.float64
        nr, d = xr.shape
        nc = xc.shape[0] if xc is not None else nr
        for i, sp in enumerate(_passes(spec)):
            if i > 0 and xc is not None and sp.ncomp == 0:
                continue                      # a noise-only pass adds nothing to a cross build (covar.py:243)
            _lib.check(self.lib.pg_kernel_build(
                self.h, _code(out.dtype), C.byref(sp), _p(hp), _p(xr), xr.stride(0), nr,
                _p(xc), xc.stride(0) if xc is not None else 0, nc, d, int(lower_only), int(i > 0),
                float(jitter) if i == 0 else 0.0, _p(out), out.stride(0), out.shape[0], out.shape[1], self._st()),
                "pg_kernel_build")
        return out

    def kernel_build_batched(self, spec, hp_all, xr, xc_all, out_all, lower_only=False, jitter=0.0):
        """out_all[e] [rows_pad, cols_pad] <- k(xr_e, xc_e) for all experts in ONE launch (pg_kernel_build_batched): hp_all [nexp, nhp];
        xr [m, d] (every expert at the same row points) or [nexp, m, d]; xc_all [nexp | 1, n, d] (one leading entry: shared points), or
        None for symmetric builds on xr.  One pg_covspec only (no accumulate passes)."""
        passes = _passes(spec)
        assert len(passes) == 1
        self._chk(hp_all, xr, xc_all, out_all)
        assert hp_all.dtype == torch.float64 and out_all.dim() == 3
        nexp = out_all.shape[0]
        nr, d = xr.shape[-2], xr.shape[-1]
        xr_stride = xr.stride(0) if (xr.dim() == 3 and xr.shape[0] > 1) else 0
        if xc_all is not None:
            nc = xc_all.shape[-2]
            xc_stride = xc_all.stride(0) if (xc_all.dim() == 3 and xc_all.shape[0] > 1) else 0
            ldc = xc_all.stride(-2)
        else:
            nc, xc_stride, ldc = nr, 0, 0
        _lib.check(self.lib.pg_kernel_build_batched(
            self.h, _code(out_all.dtype), C.byref(passes[0]), _p(hp_all), hp_all.stride(0) if hp_all.shape[0] > 1 else 0, _p(xr), xr.stride(-2),
            xr_stride, nr, _p(xc_all), ldc, xc_stride, nc, d, int(lower_only), float(jitter), _p(out_all), out_all.stride(1),
            out_all.stride(0), out_all.shape[1], out_all.shape[2], nexp, self._st()), "pg_kernel_build_batched")
        return out_all

    def kernel_grad_build(self, spec, hp, x, out):
        """out[nhp, n, n] <- dK/dtheta stack (public Covar.kernel_and_grad only)."""
        self._chk(hp, x, out)
        n, d = x.shape
        for sp in _passes(spec):              # passes write the slabs of different children
            _lib.check(self.lib.pg_kernel_grad_build(self.h, _code(out.dtype), C.byref(sp), _p(hp), _p(x), x.stride(0),
                                                     n, d, _p(out), self._st()), "pg_kernel_grad_build")
        return out

    def sqdist(self, xr, xc, out):
        """out[rows_pad, cols_pad] <- |xr_i - xc_j|^2 (xc None: xr against itself), direct differences:
        Squared_exponential.distance (covar.py:102-127) through the covariance tile kernel."""
        d = xr.shape[1]
        spec = make_spec([_lib.PG_KIND_SQDIST], [0], [])
        ones = torch.ones(d + 1, dtype=torch.float64, device=self.device)
        return self.kernel_build(spec, ones, xr, xc, out)

    # -- factorisation and solves -------------------------------------------------------------
    def potrf_worksize(self, n_pad, dtype):
        return int(self.lib.pg_potrf_worksize(_code(dtype), n_pad))

    def potrf_workspace(self, n_pad, dtype):
        return self.empty(self.potrf_worksize(n_pad, dtype), dtype=dtype)

    def potrf(self, a, invd, info):
        self._chk(a, invd, info)
        _lib.check(self.lib.pg_potrf(self.h, _code(a.dtype), a.shape[0], _p(a), a.stride(0), _p(invd), _p(info),
                                     self._st()), "pg_potrf")

    def build_factor(self, spec, hp, x, a, invd, info, minv=None, jitter=JITTER):
        """a <- k(x, x) + jitter I (lower tiles), then its Cholesky factor in place (and minv <- L^-1 if given): kernel_build
        folded into potrf / potrf_trtri -- the build of everything right of the first panel overlaps that panel's chain.  A
        Compose longer than one pg_covspec falls back to the separate calls."""
        passes = _passes(spec)
        if len(passes) != 1 or os.environ.get("PG_NO_FOLD"):
            self.kernel_build(spec, hp, x, None, a, lower_only=True, jitter=jitter)
            return self.potrf_trtri(a, invd, info, minv) if minv is not None else self.potrf(a, invd, info)
        self._chk(hp, x, a, invd, info, minv)
        assert hp.dtype == torch.float64
        n, d = x.shape
        _lib.check(self.lib.pg_build_potrf_trtri(
            self.h, _code(a.dtype), C.byref(passes[0]), _p(hp), _p(x), x.stride(0), n, d, float(jitter), _p(a), a.stride(0),
            a.shape[0], _p(invd), _p(info), _p(minv), minv.stride(0) if minv is not None else 0, self._st()),
            "pg_build_potrf_trtri")

    def build_factor_batched(self, spec, hp_all, x_all, x_stride, a_all, invd_all, info_all, minv_all=None, jitter=JITTER):
        """The same for nexp experts of one size in ONE call (pg_build_potrf_trtri_batched): hp_all [nexp, nhp], x_all [nexp | 1, n, d]
        (x_stride = 0 shares the points), a_all [nexp, n_pad, n_pad], invd_all [nexp, pg_potrf_worksize], info_all [nexp] int32,
        minv_all [nexp, n_pad, n_pad] or None.  Every launch covers all experts; the flag-coupled chain where the batch is still
        latency-bound (experts of at least 2048 points, at most 24576 rows in all), the classic chain otherwise."""
        passes = _passes(spec)
        assert len(passes) == 1
        self._chk(hp_all, x_all, a_all, invd_all, info_all, minv_all)
        assert hp_all.dtype == torch.float64 and info_all.dtype == torch.int32
        nexp, n_pad = a_all.shape[0], a_all.shape[1]
        n, d = x_all.shape[-2], x_all.shape[-1]
        _lib.check(self.lib.pg_build_potrf_trtri_batched(
            self.h, _code(a_all.dtype), C.byref(passes[0]), _p(hp_all), hp_all.stride(0), _p(x_all), x_all.stride(-2), int(x_stride),
            n, d, float(jitter), _p(a_all), a_all.stride(1), a_all.stride(0), n_pad, _p(invd_all), invd_all.stride(0), _p(info_all),
            _p(minv_all), minv_all.stride(1) if minv_all is not None else 0, minv_all.stride(0) if minv_all is not None else 0,
            nexp, self._st()), "pg_build_potrf_trtri_batched")

    def potrf_trtri_batched(self, a_all, invd_all, info_all, minv_all=None):
        """Cholesky in place (+ minv_all <- L^-1) of nexp matrices that are already in a_all [nexp, n_pad, n_pad]: the batched call
        without a folded build (X = NULL)."""
        self._chk(a_all, invd_all, info_all, minv_all)
        assert info_all.dtype == torch.int32
        nexp, n_pad = a_all.shape[0], a_all.shape[1]
        _lib.check(self.lib.pg_build_potrf_trtri_batched(
            self.h, _code(a_all.dtype), None, None, 0, None, 0, 0, n_pad, 0, 0.0, _p(a_all), a_all.stride(1), a_all.stride(0), n_pad,
            _p(invd_all), invd_all.stride(0), _p(info_all), _p(minv_all), minv_all.stride(1) if minv_all is not None else 0,
            minv_all.stride(0) if minv_all is not None else 0, nexp, self._st()), "pg_build_potrf_trtri_batched")

    def alpha_batched(self, minv_all, y_all, u_all, alpha_all, work_all):
        """alpha_e = minv_e^T (minv_e y_e) for all experts in three launches; y_all [nexp | 1, n_pad] (one row: shared targets),
        u_all [nexp, n_pad], work_all [nexp, (n_pad/256) n_pad]."""
        self._chk(minv_all, y_all, u_all, alpha_all, work_all)
        _lib.check(self.lib.pg_alpha_batched(self.h, _code(minv_all.dtype), minv_all.shape[1], _p(minv_all), minv_all.stride(1),
                                             minv_all.stride(0), _p(y_all), y_all.stride(0) if y_all.shape[0] > 1 else 0, _p(u_all), u_all.stride(0), _p(alpha_all),
                                             alpha_all.stride(0), _p(work_all), work_all.stride(0), minv_all.shape[0], self._st()),
                   "pg_alpha_batched")

    def alpha_nlml_batched(self, minv_all, y_all, u_all, alpha_all, work_all, n, out_all):
        """alpha_e and NLML_e for all experts (pg_alpha_nlml_batched): out_all [nexp, >= 1] float64 receives NLML_e in column 0."""
        self._chk(minv_all, y_all, u_all, alpha_all, work_all, None)
        assert out_all.dtype == torch.float64 and out_all.is_cuda
        _lib.check(self.lib.pg_alpha_nlml_batched(
            self.h, _code(minv_all.dtype), int(n), minv_all.shape[1], _p(minv_all), minv_all.stride(1), minv_all.stride(0), _p(y_all),
            y_all.stride(0) if y_all.shape[0] > 1 else 0, _p(u_all), u_all.stride(0), _p(alpha_all), alpha_all.stride(0), _p(work_all),
            work_all.stride(0), _p(out_all), out_all.stride(0), minv_all.shape[0], self._st()), "pg_alpha_nlml_batched")

    def lauum_batched(self, minv_all, kinv_all):
        self._chk(minv_all, kinv_all)
        _lib.check(self.lib.pg_lauum_batched(self.h, _code(minv_all.dtype), minv_all.shape[1], _p(minv_all), minv_all.stride(1),
                                             minv_all.stride(0), _p(kinv_all), kinv_all.stride(1), kinv_all.stride(0), minv_all.shape[0],
                                             self._st()), "pg_lauum_batched")

    def nlml_grad_batched(self, spec, hp_all, x_all, x_stride, n, kinv_all, alpha_all, grad_all, work):
        """grad_all [nexp, >= nhp] (a view with row stride: e.g. outs[:, 1:]) <- every expert's gradient in two launches."""
        passes = _passes(spec)
        self._chk(hp_all, x_all, kinv_all, alpha_all, work)
        assert grad_all.dtype == torch.float64 and grad_all.is_cuda and grad_all.stride(-1) == 1
        nexp, nhp = kinv_all.shape[0], hp_all.shape[-1]
        for sp in passes:
            _lib.check(self.lib.pg_nlml_grad_batched(
                self.h, _code(kinv_all.dtype), C.byref(sp), _p(hp_all), hp_all.stride(0), _p(x_all), x_all.stride(-2), int(x_stride), int(n),
                x_all.shape[-1], _p(kinv_all), kinv_all.stride(1), kinv_all.stride(0), _p(alpha_all), alpha_all.stride(0), _p(grad_all),
                grad_all.stride(0), nhp, _p(work), work.numel(), nexp, self._st()), "pg_nlml_grad_batched")

    def potrf_trtri(self, a, invd, info, minv):
        """Cholesky in place + minv = L^-1, fused so that part of the inverse overlaps the factorisation's tail."""
        self._chk(a, invd, info, minv)
        _lib.check(self.lib.pg_potrf_trtri(self.h, _code(a.dtype), a.shape[0], _p(a), a.stride(0), _p(invd), _p(info),
                                           _p(minv), minv.stride(0), self._st()), "pg_potrf_trtri")

    def potrs_vec(self, chol, invd, y, x, work=None):
        if work is None:
            work = self.empty(self.lib.pg_potrs_vec_worksize(_code(chol.dtype), chol.shape[0]), dtype=chol.dtype)
        self._chk(chol, invd, y, x, work)
        _lib.check(self.lib.pg_potrs_vec(self.h, _code(chol.dtype), chol.shape[0], _p(chol), chol.stride(0), _p(invd),
                                         _p(y), _p(x), _p(work), self._st()), "pg_potrs_vec")

    def potrs(self, chol, invd, b, minv=None, triangular_only=False):
        """x = K^-1 b (or L^-1 b) for a matrix right-hand side b [n_pad, nrhs_pad] (nrhs_pad a multiple of 128): pg_potrs / pg_trsm_lower."""
        n, nrhs = b.shape
        x = self.empty(n, nrhs, dtype=b.dtype)
        work = self.empty(self.lib.pg_potrs_worksize(_code(b.dtype), n, nrhs, int(minv is not None)), dtype=b.dtype)
        self._chk(chol, invd, b, minv, x, work)
        fn = self.lib.pg_trsm_lower if triangular_only else self.lib.pg_potrs
        _lib.check(fn(self.h, _code(b.dtype), n, nrhs, _p(chol), chol.stride(0) if chol is not None else 0, _p(invd), _p(minv),
                      minv.stride(0) if minv is not None else 0, _p(b), b.stride(0), _p(x), x.stride(0), _p(work), self._st()),
                   "pg_trsm_lower" if triangular_only else "pg_potrs")
        return x

    def potri(self, chol, invd, kinv, work=None):
        """kinv(lower) = K^-1 from the factor (pg_trtri + pg_lauum in one call)."""
        n = chol.shape[0]
        if work is None:
            work = self.empty(n, n, dtype=chol.dtype)
        self._chk(chol, invd, kinv, work)
        _lib.check(self.lib.pg_potri(self.h, _code(chol.dtype), n, _p(chol), chol.stride(0), _p(invd), _p(kinv), kinv.stride(0),
                                     _p(work), self._st()), "pg_potri")

    def logdet(self, chol, n, out):
        self._chk(chol, out)
        _lib.check(self.lib.pg_logdet(self.h, _code(chol.dtype), int(n), _p(chol), chol.stride(0), _p(out), self._st()), "pg_logdet")

    def trtri(self, chol, invd, minv):
        self._chk(chol, invd, minv)
        _lib.check(self.lib.pg_trtri(self.h, _code(chol.dtype), chol.shape[0], _p(chol), chol.stride(0), _p(invd),
                                     _p(minv), minv.stride(0), self._st()), "pg_trtri")

    def lauum(self, minv, kinv):
        self._chk(minv, kinv)
        _lib.check(self.lib.pg_lauum(self.h, _code(minv.dtype), minv.shape[0], _p(minv), minv.stride(0), _p(kinv),
                                     kinv.stride(0), self._st()), "pg_lauum")

    def trmv(self, minv, x, y, trans, work=None):
        self._chk(minv, x, y, work)
        _lib.check(self.lib.pg_trmv(self.h, _code(minv.dtype), minv.shape[0], _p(minv), minv.stride(0), int(trans),
                                    _p(x), _p(y), _p(work), self._st()), "pg_trmv")

    def tril(self, a, n):
        self._chk(a)
        _lib.check(self.lib.pg_tril(self.h, _code(a.dtype), n, _p(a), a.stride(0), self._st()), "pg_tril")

    # -- NLML ---------------------------------------------------------------------------------
    def nlml_value(self, chol, y, alpha, n, out):
        self._chk(chol, y, alpha, out)
        _lib.check(self.lib.pg_nlml_value(self.h, _code(chol.dtype), n, _p(chol), chol.stride(0), _p(y), _p(alpha),
                                          _p(out), self._st()), "pg_nlml_value")

    def alpha_nlml_async(self, chol, minv, y, u, alpha, work, n, out):
        """alpha = minv^T (minv y) and out[0] = NLML (out[1]: log det, scratch), overlapped with the next pg_lauum."""
        self._chk(chol, minv, y, u, alpha, work, out)
        assert out.dtype == torch.float64 and out.numel() >= 2
        _lib.check(self.lib.pg_alpha_nlml_async(self.h, _code(chol.dtype), int(n), chol.shape[0], _p(chol), chol.stride(0), _p(minv),
                                                minv.stride(0), _p(y), _p(u), _p(alpha), _p(work), _p(out), self._st()),
                   "pg_alpha_nlml_async")

    def nlml_grad_worksize(self, n, nhp):
        return self.lib.pg_nlml_grad_worksize(n, nhp)

    def nlml_grad(self, spec, hp, x, n, kinv, alpha, grad, work):
        self._chk(hp, x, kinv, alpha, grad, work)
        for sp in _passes(spec):              # each pass fills the gradient entries of its own children
            _lib.check(self.lib.pg_nlml_grad(self.h, _code(kinv.dtype), C.byref(sp), _p(hp), _p(x), x.stride(0), n,
                                             x.shape[1], _p(kinv), kinv.stride(0), _p(alpha), _p(grad), grad.numel(),
                                             _p(work), work.numel(), self._st()), "pg_nlml_grad")

    # -- prediction ---------------------------------------------------------------------------
    def predict_mean_q(self, ks, minv, alpha, mean, var, kss, work):
        """mean = Ks^T alpha; var = kss - colsum((Minv Ks)^2) (var None: mean only)."""
        q = var
        self._chk(ks, minv, alpha, mean, q, work)
        _lib.check(self.lib.pg_predict_mean_q(self.h, _code(ks.dtype), ks.shape[0], ks.shape[1], _p(ks), ks.stride(0),
                                              _p(minv), minv.stride(0) if minv is not None else 0, _p(alpha),
                                              _p(mean), _p(q), float(kss), _p(work), self._st()),
                   "pg_predict_mean_q")

    def predict_mean_q_kt(self, kt, minv, alpha, mean, var, kss, work):
        """The same from kt [m_pad, n_pad] = k(xp, x) (test-point-major): mean = Kt alpha; var = kss - colsum((Minv Kt^T)^2)."""
        q = var
        self._chk(kt, minv, alpha, mean, q, work)
        _lib.check(self.lib.pg_predict_mean_q_kt(self.h, _code(kt.dtype), kt.shape[1], kt.shape[0], _p(kt), kt.stride(0),
                                                 _p(minv), minv.stride(0) if minv is not None else 0, _p(alpha),
                                                 _p(mean), _p(q), float(kss), _p(work), self._st()),
                   "pg_predict_mean_q_kt")

    def predict_mean_q_kt_batched(self, kt_all, minv_all, alpha_all, mean_all, var_all, spec, hp_all, work_all):
        """All experts' means (and diagonal variances, var_all not None) in three launches (pg_predict_mean_q_kt_batched): kt_all
        [nexp, m_pad, n_pad], minv_all [nexp, n_pad, n_pad] (any common stride), alpha_all [nexp, n_pad], mean_all / var_all [nexp, m_pad],
        hp_all [nexp | 1, nhp], work_all [nexp, (n_pad/64) m_pad].  kss is formed on the device from hp_all."""
        passes = _passes(spec)
        assert len(passes) == 1
        self._chk(kt_all, alpha_all, hp_all, work_all)
        nexp, m_pad, n_pad = kt_all.shape
        want_var = var_all is not None
        for t in (mean_all, var_all):          # rows may be slices of longer rows (chunks of test points): unit stride inside a row
            assert t is None or (t.is_cuda and t.dim() == 2 and t.stride(1) == 1 and t.shape[1] >= m_pad and (nexp == 1 or t.stride(0) >= m_pad))
        if want_var:
            assert minv_all.is_cuda and minv_all.stride(-1) == 1 and hp_all.dtype == torch.float64
        _lib.check(self.lib.pg_predict_mean_q_kt_batched(
            self.h, _code(kt_all.dtype), n_pad, m_pad, _p(kt_all), kt_all.stride(1), kt_all.stride(0),
            _p(minv_all) if want_var else None, minv_all.stride(-2) if want_var else 0, minv_all.stride(0) if want_var else 0,
            _p(alpha_all), alpha_all.stride(0), _p(mean_all), mean_all.stride(0), _p(var_all), var_all.stride(0) if want_var else 0,
            C.byref(passes[0]), _p(hp_all), hp_all.stride(0) if hp_all.shape[0] > 1 else 0, _p(work_all), work_all.stride(0), nexp,
            self._st()), "pg_predict_mean_q_kt_batched")

    def trmm_lower(self, minv, ks, v):
        self._chk(minv, ks, v)
        _lib.check(self.lib.pg_trmm_lower(self.h, _code(ks.dtype), ks.shape[0], ks.shape[1], _p(minv), minv.stride(0),
                                          _p(ks), ks.stride(0), _p(v), v.stride(0), self._st()), "pg_trmm_lower")

    def syrk_tn_sub(self, v, c, lower_only=True):
        self._chk(v, c)
        _lib.check(self.lib.pg_syrk_tn_sub(self.h, _code(v.dtype), c.shape[0], v.shape[0], _p(v), v.stride(0), _p(c),
                                           c.stride(0), int(lower_only), self._st()), "pg_syrk_tn_sub")

    def trmm_lower_kt(self, minv, kt, vt):
        """vt = kt minv^T (= (minv ks)^T) from the test-point-major cross-covariance: kt, vt [m_pad, n_pad] or [nexp, m_pad, n_pad];
        minv [n_pad, n_pad], [nexp, n_pad, n_pad], or a list of nexp matrices that sit at one stride in memory (views of a stack)."""
        if isinstance(minv, (list, tuple)):
            self._chk(kt, vt, *minv)
            m0 = minv[0]
            step = (minv[1].data_ptr() - m0.data_ptr()) // m0.element_size() if len(minv) > 1 else 0
            assert all(mi.data_ptr() - m0.data_ptr() == i * step * m0.element_size() and mi.stride(0) == m0.stride(0)
                       for i, mi in enumerate(minv)), "the experts' inverses do not sit at one stride"
            ldm = m0.stride(0)
        else:
            self._chk(minv, kt, vt)
            m0, ldm, step = minv, minv.stride(-2), (minv.stride(0) if minv.dim() == 3 else 0)
        nexp = kt.shape[0] if kt.dim() == 3 else 1
        _lib.check(self.lib.pg_trmm_lower_kt_batched(
            self.h, _code(kt.dtype), kt.shape[-1], kt.shape[-2], _p(m0), ldm, step, _p(kt), kt.stride(-2),
            kt.stride(0) if kt.dim() == 3 else 0, _p(vt), vt.stride(-2), vt.stride(0) if vt.dim() == 3 else 0, nexp, self._st()),
            "pg_trmm_lower_kt_batched")

    def syrk_nt_sub_batched(self, vt_all, c_all, lower_only=True):
        """c_all[e] -= vt_all[e] vt_all[e]^T for all experts in one launch: vt_all [nexp, m_pad, n_pad], c_all [nexp, m_pad, m_pad]."""
        self._chk(vt_all, c_all)
        assert vt_all.dim() == 3 and c_all.dim() == 3 and vt_all.shape[0] == c_all.shape[0]
        _lib.check(self.lib.pg_syrk_nt_sub_batched(self.h, _code(c_all.dtype), c_all.shape[1], vt_all.shape[2], _p(vt_all),
                                                   vt_all.stride(1), vt_all.stride(0), _p(c_all), c_all.stride(1), c_all.stride(0),
                                                   c_all.shape[0], int(lower_only), self._st()), "pg_syrk_nt_sub_batched")

    # -- grBCM --------------------------------------------------------------------------------
    def grbcm_local_terms(self, mean_c, var_c, var_g, is_first, accumulate, out, beta=None, prec=None):
        self._chk(mean_c, var_c, var_g, out, beta, prec)
        assert out.dtype == torch.float64 and out.shape[0] == 3
        _lib.check(self.lib.pg_grbcm_local_terms(self.h, _code(mean_c.dtype), mean_c.numel(), _p(mean_c), _p(var_c),
                                                 _p(var_g), int(is_first), int(accumulate), _p(out), out.stride(0),
                                                 _p(beta), _p(prec), self._st()), "pg_grbcm_local_terms")

    def grbcm_local_terms_batched(self, mean_all, var_all, var_g, first, accumulate, out, beta=None, prec=None):
        """The terms of all owned experts in one launch: mean_all / var_all [nexp, >= m] (row stride arbitrary), out [3, m] float64,
        beta / prec [nexp, m] float64 views (common row stride) or None; `first`: index of the committee's first expert among these, -1: none."""
        self._chk(var_g, out)
        assert out.dtype == torch.float64 and out.shape[0] == 3 and mean_all.stride(-1) == 1 and var_all.stride(-1) == 1
        m = var_g.numel()
        nexp = mean_all.shape[0]
        if beta is not None:
            assert beta.stride(-1) == 1 and prec.stride(-1) == 1 and (nexp == 1 or beta.stride(0) == prec.stride(0))
        _lib.check(self.lib.pg_grbcm_local_terms_batched(
            self.h, _code(mean_all.dtype), m, _p(mean_all), mean_all.stride(0), _p(var_all), var_all.stride(0), _p(var_g), nexp, int(first),
            int(accumulate), _p(out), out.stride(0), _p(beta), _p(prec), beta.stride(0) if beta is not None else 0, self._st()),
            "pg_grbcm_local_terms_batched")

    def grbcm_finish(self, sums, mean_g, var_g, mean, var, beta0=None, prec0=None):
        self._chk(sums, mean_g, var_g, mean, var, beta0, prec0)
        _lib.check(self.lib.pg_grbcm_finish(self.h, _code(mean_g.dtype), mean_g.numel(), _p(sums), sums.stride(0),
                                            _p(mean_g), _p(var_g), _p(mean), _p(var), _p(beta0), _p(prec0),
                                            self._st()), "pg_grbcm_finish")

    def grbcm_weighted_prec(self, prec, beta, acc, m, accumulate):
        self._chk(prec, beta, acc)
        _lib.check(self.lib.pg_grbcm_weighted_prec(self.h, _code(acc.dtype), m, acc.shape[0], _p(prec), prec.stride(0), _p(beta),
                                                   _p(acc), acc.stride(0), int(accumulate), self._st()), "pg_grbcm_weighted_prec")

    def symmetrize(self, a, n):
        self._chk(a)
        _lib.check(self.lib.pg_symmetrize(self.h, _code(a.dtype), n, _p(a), a.stride(0), self._st()), "pg_symmetrize")

    def grbcm_finish_full(self, sums, mean_g, var_g, cov, mean):
        self._chk(sums, mean_g, var_g, cov, mean)
        _lib.check(self.lib.pg_grbcm_finish_full(self.h, _code(mean_g.dtype), mean_g.numel(), _p(sums), sums.stride(0), _p(mean_g),
                                                 _p(var_g), _p(cov), cov.stride(0), _p(mean), self._st()), "pg_grbcm_finish_full")

    def spd_inverse_lower(self, a_pad):
        """a_pad (padded SPD, lower triangle valid) -> its inverse's lower triangle in a new buffer; raises on a bad pivot."""
        n = a_pad.shape[0]
        invd = self.potrf_workspace(n, a_pad.dtype)
        info = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.potrf(a_pad, invd, info)
        minv = self.empty(n, n, dtype=a_pad.dtype)
        self.trtri(a_pad, invd, minv)
        out = self.empty(n, n, dtype=a_pad.dtype)
        self.lauum(minv, out)
        return out, info

    def spd_inverse_lower_batched(self, a_all):
        """a_all [nexp, n_pad, n_pad] (padded SPD, lower triangles valid; overwritten) -> (the inverses' lower triangles IN a_all,
        info [nexp]): every step of factor, L^-1 and L^-T L^-1 is one launch over all matrices."""
        nexp, n = a_all.shape[0], a_all.shape[1]
        invd = self.empty(nexp, self.potrf_worksize(n, a_all.dtype), dtype=a_all.dtype)
        info = torch.zeros(nexp, dtype=torch.int32, device=self.device)
        minv = self.empty(nexp, n, n, dtype=a_all.dtype)
        self.potrf_trtri_batched(a_all, invd, info, minv)
        self.lauum_batched(minv, a_all)
        return a_all, info

    def sqdist_argmin(self, x, centres, dist=None, idx=None):
        """dist[n, m] = squared distances, idx[n] (int32) = nearest centre; either output may be None."""
        self._chk(x, centres, dist, idx)
        _lib.check(self.lib.pg_sqdist_argmin(self.h, _code(x.dtype), _p(x), x.stride(0), x.shape[0], _p(centres),
                                             centres.stride(0), centres.shape[0], x.shape[1], _p(dist),
                                             dist.stride(0) if dist is not None else 0, _p(idx), self._st()), "pg_sqdist_argmin")

    # -- raw GEMM core (tests, roofline micro-benchmark) ---------------------------------------
    def gemm_raw(self, variant, m, n, k, alpha, a, b, beta, c, tri=0, klo=0, khi=0):
        self._chk(a, b, c)
        _lib.check(self.lib.pg_gemm_raw(self.h, _code(c.dtype), variant, m, n, k, float(alpha), _p(a), a.stride(0),
                                        _p(b), b.stride(0), float(beta), _p(c), c.stride(0), tri, klo, khi,
                                        self._st()), "pg_gemm_raw")

    def set_lookahead(self, on):
        _lib.check(self.lib.pg_set_lookahead(self.h, int(on)), "pg_set_lookahead")

    def set_outer_panel(self, columns):
        _lib.check(self.lib.pg_set_outer_panel(self.h, int(columns)), "pg_set_outer_panel")

    def set_recursive_split(self, min_n):
        """From min_n points on the fused factor-and-invert call splits the matrix recursively (0: never; default 16384)."""
        _lib.check(self.lib.pg_set_recursive_split(self.h, int(min_n)), "pg_set_recursive_split")

    def set_coupled_chain(self, on):
        _lib.check(self.lib.pg_set_coupled_chain(self.h, int(on)), "pg_set_coupled_chain")

    def set_spin_budget(self, microseconds):
        """Wall-time bound of one wait of the coupled chain; 0: scaled to the call (default); < 0: every wait expires at once
        (test hook of the fall-back)."""
        _lib.check(self.lib.pg_set_spin_budget(self.h, int(microseconds)), "pg_set_spin_budget")

    def chain_timeouts(self):
        return int(self.lib.pg_chain_timeouts(self.h))

    def set_rearm_after(self, calls):
        """After a time-out the handle takes the coupled chain back by itself once this many further factorisations have been
        enqueued (default 8; 0: never)."""
        _lib.check(self.lib.pg_set_rearm_after(self.h, int(calls)), "pg_set_rearm_after")

    def chain_rearms(self):
        return int(self.lib.pg_chain_rearms(self.h))

    def wait_budget_us(self, n):
        return int(self.lib.pg_wait_budget_us(self.h, int(n)))

    def spin_probe(self, n):
        """One bounded wait on a flag nobody sets, with the budget of an n x n factorisation: (milliseconds waited, flag came)."""
        scratch = torch.zeros(4, dtype=torch.int64, device=self.device)
        _lib.check(self.lib.pg_spin_probe(self.h, int(n), _p(scratch), self._st()), "pg_spin_probe")
        torch.cuda.synchronize()
        v = scratch.tolist()
        return v[2] * 1e-5, bool(v[3])

    def recover_from_timeout(self):
        """A factorisation reported info = -1 (a wait of the coupled chain expired: include/pygpr_hip.h).  The library has
        switched the handle to the classic chain by itself (pinned word polled at every entry point); make sure, count, and
        tell the caller to repeat its sequence from the covariance build."""
        torch.cuda.synchronize()
        self.chain_timeouts()                      # polls the pinned word
        if self.coupled_chain():                   # (-1: off as after a time-out -- the rows stream stays, the handle re-arms itself)
            _lib.check(self.lib.pg_set_coupled_chain(self.h, -1), "pg_set_coupled_chain")
        self.fallbacks = getattr(self, "fallbacks", 0) + 1

    def build_factor_checked(self, spec, hp, x, a, invd, info, minv=None, jitter=JITTER):
        """Blocking build + factor with the fall-back inside the call (pg_build_potrf_trtri_checked); returns info."""
        passes = _passes(spec)
        assert len(passes) == 1
        self._chk(hp, x, a, invd, info, minv)
        n, d = x.shape
        out = C.c_int(0)
        _lib.check(self.lib.pg_build_potrf_trtri_checked(
            self.h, _code(a.dtype), C.byref(passes[0]), _p(hp), _p(x), x.stride(0), n, d, float(jitter), _p(a), a.stride(0),
            a.shape[0], _p(invd), _p(info), _p(minv), minv.stride(0) if minv is not None else 0, self._st(), C.byref(out)),
            "pg_build_potrf_trtri_checked")
        return out.value

    def coupled_chain(self):
        return int(self.lib.pg_coupled_chain(self.h))

    def last_coupled_panels(self):
        return int(self.lib.pg_last_coupled_panels(self.h))

    def set_deferred_block(self, on):
        _lib.check(self.lib.pg_set_deferred_block(self.h, int(on)), "pg_set_deferred_block")

    def last_deferred_panels(self):
        return int(self.lib.pg_last_deferred_panels(self.h))

    def leaf_raw(self, a, inv, info, ablate=0):
        self._chk(a, inv, info)
        _lib.check(self.lib.pg_leaf_raw(self.h, _code(a.dtype), _p(a), a.stride(0), _p(inv), inv.stride(0) if inv is not None else 0,
                                        _p(info), int(ablate), self._st()), "pg_leaf_raw")

    def profile(self, on):
        _lib.check(self.lib.pg_profile(self.h, int(on)), "pg_profile")

    def profile_read(self):
        f, ms, n = C.c_double(), C.c_double(), C.c_long()
        _lib.check(self.lib.pg_profile_read(self.h, C.byref(f), C.byref(ms), C.byref(n)), "pg_profile_read")
        return f.value, ms.value, n.value


_OPS = None


def get_ops():
    """The process-wide device-op object; raises (never falls back) without library + GPU."""
    global _OPS
    if _OPS is None:
        _OPS = HipOps()
    return _OPS


def _passes(spec):
    return spec if isinstance(spec, (list, tuple)) else [spec]


def make_specs(kinds, offs, noise_offs):
    """The passes of a Compose of any length: pg_covspec holds PG_MAX_COMP stationary and PG_MAX_COMP noise children, a
    longer sum (the reference's Compose is unlimited, covar.py:28-81) is evaluated PG_MAX_COMP children at a time."""
    q = _lib.PG_MAX_COMP
    npass = max(1, -(-len(kinds) // q), -(-len(noise_offs) // q))
    return [make_spec(kinds[i * q: (i + 1) * q], offs[i * q: (i + 1) * q], noise_offs[i * q: (i + 1) * q]) for i in range(npass)]


def make_spec(kinds, offs, noise_offs):
    s = CovSpec()
    if len(kinds) > _lib.PG_MAX_COMP or len(noise_offs) > _lib.PG_MAX_COMP:
        raise ValueError("one pg_covspec holds at most %d stationary and %d noise kernels (make_specs splits a longer Compose)"
                         % (_lib.PG_MAX_COMP, _lib.PG_MAX_COMP))
    s.ncomp = len(kinds)
    for i, (k, o) in enumerate(zip(kinds, offs)):
        s.kind[i], s.off[i] = k, o
    s.nnoise = len(noise_offs)
    for i, o in enumerate(noise_offs):
        s.noise_off[i] = o
    return s
