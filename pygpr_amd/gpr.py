"""GP models with PyGPR's class surface (reference: PyGPR/gpr.py) on the MI355X.

`Exact_GP.update` = covariance build + 1e-7 jitter + Cholesky + alpha = K^-1 y (gpr.py:65-74);
`predict` = K* build, mean K* alpha, and either the diagonal predictive variance
diag(K**) - rowsum(K* o (K^-1 K*^T)^T) (gpr.py:96-106) or the full covariance (gpr.py:108-120).
All state lives on the GPU in padded buffers (n rounded up to 256, identity in the padding); the
leading expert dimension of the reference's batched models lives in stacked tensors (`_batch`): factorisation, inverse,
weights AND the prediction of all experts are batched calls (round 5: K* of every expert in one launch, mean + diagonal
variance in three, gpr.py:76-106 on x [nc, n, d]; PG_PREDICT_SERIAL=1 walks the experts one by one as rounds 1-4 did).

Differences a caller can observe:
  * the triangular solves against K* use the explicit inverse factor L^-1 (cached per update), so the
    variance is a GEMM with a fused column-sum-of-squares epilogue instead of `cholesky_solve`
  * K** is never built for var="diag": its diagonal is sum sigma_c^2 + sum sigma_n^2 analytically
  * assigning `model.x` / `model.y` marks the model dirty (the reference keeps a stale factor there)
  * `krn`, `krnchd`, `wt` are read-only views materialised on access
"""
import os
from typing import Sequence

import torch
from torch import Tensor

from ._ops import JITTER, get_ops, pad_to
from .covar import Covar, layout, spec_of

_CHUNK = 8192  # test points per device batch
# Experts of a batched model are factorised in ONE batched call (every launch covers all experts: pg_build_potrf_trtri_batched)
# up to this padded size; above it the per-step launches no longer matter and each expert takes the single-model schedule with
# its flag-coupled chain, one after the other.  PG_BATCH_MAX_N overrides (0: never batch).
_BATCH_MAX_N = int(os.environ.get("PG_BATCH_MAX_N", "12288"))
_FULL_VT_BYTES = 16 << 30   # predict(var="full"): experts whose V^T fit this together share one rank-n update launch (and at most
                            # 40 % of the device memory that is free when the call starts: `_group_budget`)
_BATCH_EAGER_N = 4096   # batched experts up to this size form L^-1 with the factor even when nobody asked for variances: the
                        # batched inverse + three batched mat-vec launches are cheaper than one substitution sweep per expert


class ChainTimeout(torch.linalg.LinAlgError):
    """info = -1: a bounded wait inside the factorisation's flag-coupled chain expired -- kernels of the library's streams did
    not run concurrently (include/pygpr_hip.h).  Not a property of the matrix.  Every caller in this package answers it by
    repeating the evaluation on the classic chain (`HipOps.recover_from_timeout`); the exception only escapes if the repeat
    fails too.  A LinAlgError subclass so that the committee's status-word paths (gr_bcm.py) treat it like any failed expert."""


def _lin_alg_error(info: int, note: str = ""):
    if int(info) < 0:
        err = ChainTimeout("pygpr_amd: the factorisation's coupled chain timed out twice (info = %d): kernels of different "
                           "streams do not run concurrently in this environment%s" % (int(info), note))
        err.pg_info = int(info)
        return err
    err = torch.linalg.LinAlgError(
        "cholesky: The factorization could not be completed because the input is not positive-definite "
        "(the leading minor of order %d is not positive-definite)%s." % (info, note))
    err.pg_info = int(info)
    return err


def _checked(enqueue, infos):
    """Run `enqueue()` (device work that ends in the factorisation status words `infos()` reads after one sync).  info < 0 is
    the coupled chain's time-out: switch the handle to the classic chain and repeat ONCE from the start -- tc.cholesky
    (gpr.py:69) never fails on a positive-definite matrix.  Returns the final status words."""
    enqueue()
    vals = [int(v) for v in infos()]
    if any(v < 0 for v in vals):
        get_ops().recover_from_timeout()
        enqueue()
        vals = [int(v) for v in infos()]
    return vals


def _group_budget(cap, need=None):
    """Bytes a prediction may spend on the per-expert scratch of ONE launch group: `cap`, but never more than 40 % of what the device
    has free right now (torch's cached blocks count as free: they are re-used) -- a fuller or smaller device gets smaller groups,
    down to one expert per launch, instead of an out-of-memory error.  `need`: what the caller wants in all; up to 1 GiB is granted
    without asking (torch's memory statistics cost 0.1 ms per query -- 40 % of a prediction at the reference's test sizes)."""
    if need is not None and need <= (1 << 30):
        return cap
    try:
        free, _ = torch.cuda.mem_get_info()
        free += torch.cuda.memory_reserved() - torch.cuda.memory_allocated()
    except Exception:
        return cap
    return max(1, min(cap, int(0.4 * free)))


def _stacked_rows(ts):
    """ts: per-expert 1-D device tensors.  If they are rows of one buffer at a constant stride, that buffer as a [len(ts), m] view;
    None otherwise (the committee's batched aggregation kernel takes a base pointer and a stride)."""
    if not ts:
        return None
    t0 = ts[0]
    if any(t.dim() != 1 or t.stride(0) != 1 or t.dtype != t0.dtype or t.numel() != t0.numel() for t in ts):
        return None
    base = t0.untyped_storage().data_ptr()
    if any(t.untyped_storage().data_ptr() != base for t in ts):      # views of ONE allocation only
        return None
    item = t0.element_size()
    step = (ts[1].data_ptr() - t0.data_ptr()) // item if len(ts) > 1 else t0.numel()
    if len(ts) > 1 and (step < t0.numel() or any(t.data_ptr() - t0.data_ptr() != i * step * item for i, t in enumerate(ts))):
        return None
    return torch.as_strided(t0, (len(ts), t0.numel()), (step, 1))


def _stacked_pair(a, b):
    """a, b: [nc, m] views of the same shape and strides inside ONE allocation, b behind a: the [2, nc, m] view over both; else None."""
    if a.shape != b.shape or a.stride() != b.stride() or a.dtype != b.dtype:
        return None
    if a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr():
        return None
    step = (b.data_ptr() - a.data_ptr()) // a.element_size()
    if step <= 0 or step < (a.shape[0] - 1) * a.stride(0) + a.shape[1]:
        return None
    return torch.as_strided(a, (2,) + tuple(a.shape), (step,) + tuple(a.stride()))


class GPR:
    """Base class for Gaussian process regression models (PyGPR/gpr.py:13-43)."""

    def __init__(self, x: Tensor, y: Tensor, cov: Covar) -> None:
        self._x: Tensor = x
        self._y: Tensor = y
        self.cov = cov
        self.params: Tensor = NotImplemented
        self.need_upd: bool = True
        return None

    @property
    def x(self) -> Tensor:
        return self._x

    @x.setter
    def x(self, value: Tensor) -> None:
        self._x = value
        self._data_changed()

    @property
    def y(self) -> Tensor:
        return self._y

    @y.setter
    def y(self, value: Tensor) -> None:
        self._y = value
        self._data_changed()

    def _data_changed(self) -> None:
        self.need_upd = True

    def set_params(self, params: Tensor) -> None:
        self.params = torch.clone(params)
        self.need_upd = True
        return None

    def update(self) -> None:
        raise NotImplementedError

    def predict(self, xp: Tensor, var: str) -> Sequence[Tensor]:
        raise NotImplementedError

    def predict_var(self, xp: Tensor, **kwrgs: Tensor) -> Tensor:
        raise NotImplementedError

    def predict_covar(self, xp: Tensor, **kwargs: Tensor) -> Tensor:
        raise NotImplementedError


class _Data:
    """Device copy of one (x, y) pair: x [n, d], y zero-padded to n_pad."""

    __slots__ = ("x", "y", "n", "n_pad")


class _Expert:
    """Device state of one expert (its data may be shared with other experts: batched params on unbatched x)."""

    __slots__ = ("x", "y", "n", "n_pad", "chol", "invd", "alpha", "minv", "minv_valid", "work", "hp", "info")

    def __init__(self, data):
        self.x, self.y, self.n, self.n_pad = data.x, data.y, data.n, data.n_pad
        self.chol = self.invd = self.alpha = self.minv = self.work = self.hp = self.info = None
        self.minv_valid = False


class Exact_GP(GPR):
    """Exact GP model (PyGPR/gpr.py:46-120); x [(nc), n, d], y [(nc), n], params [(nc), nhp]."""

    def __init__(self, x: Tensor, y: Tensor, cov: Covar, eager_inverse: bool = False) -> None:
        super().__init__(x, y, cov)
        self.params: Tensor = cov.init_params(x)
        self._experts = None
        self._data = None
        self._data_key = None
        self._bat = None
        self._pbuf = None
        self._x_all = self._y_all = None
        self.need_upd: bool = True
        # eager_inverse: form L^-1 inside update() (fused with the Cholesky) and take alpha = L^-T (L^-1 y) from two
        # triangular mat-vecs.  Worth it whenever predictive variances follow (grBCM experts); wasted work otherwise.
        self.eager_inverse = eager_inverse
        return None

    # ---- device residency -------------------------------------------------------------------
    @property
    def dtype(self):
        return self._x.dtype if self._x.dtype == torch.float32 else torch.float64

    @property
    def batched(self) -> bool:
        """More than one expert: the reference's kernels squeeze a batch of one away (covar.py:161-165), so x [1, n, d]
        behaves like x [n, d]."""
        return len(self._device_experts()) > 1

    def _data_changed(self) -> None:
        self._experts = None
        self._data = None
        self._pbuf = None
        self.need_upd = True

    def _device_data(self):
        """x / y on the device, one _Data per leading index of x, uploaded once per (tensor identity, version): in-place
        edits of model.x / model.y (`model.y.copy_(new)`) are seen by the next evaluation, as in the reference, which
        re-reads model.x / model.y on every call (loss.py:37,43)."""
        key = (id(self._x), self._x._version, id(self._y), self._y._version)
        if self._data is None or self._data_key != key:
            ops = get_ops()
            xb = self._x.reshape(-1, self._x.shape[-2], self._x.shape[-1])
            yb = self._y.reshape(-1, self._y.shape[-1])
            nbd = max(xb.shape[0], yb.shape[0])
            if xb.shape[0] not in (1, nbd) or yb.shape[0] not in (1, nbd):
                raise RuntimeError("batch dimensions of x and y do not broadcast")
            x_all = ops.to_device(xb, self.dtype)                       # [nbx, n, d]: ONE upload; experts take views
            n_pad = pad_to(xb.shape[1])
            y_all = ops.zeros(nbd, n_pad, dtype=self.dtype)
            y_all[:, : xb.shape[1]] = ops.to_device(yb, self.dtype).expand(nbd, -1) if yb.shape[0] != nbd else ops.to_device(yb, self.dtype)
            data = []
            for b in range(nbd):
                dt = _Data()
                dt.n = xb.shape[1]
                dt.n_pad = n_pad
                dt.x = x_all[b % x_all.shape[0]]
                dt.y = y_all[b]
                data.append(dt)
            self._x_all, self._y_all = x_all, y_all
            self._data, self._data_key = data, key
            self._experts = None
        return self._data

    def _device_experts(self):
        """One _Expert per model of the batch: max(batch of x / y, batch of params) of them, as cov.kernel(params, x)
        broadcasts in the reference (gpr.py:67; params [nc, nhp] on an unbatched x are nc models on the same points)."""
        data = self._device_data()
        nb = max(len(data), self._hp_rows().shape[0])
        if len(data) not in (1, nb) or self._hp_rows().shape[0] not in (1, nb):
            raise RuntimeError("batch dimensions of params and x do not broadcast")
        if self._experts is None or len(self._experts) != nb:
            self._experts = [_Expert(data[b % len(data)]) for b in range(nb)]
            self._bat = None
            self.need_upd = True
        return self._experts

    def _batch(self):
        """Stacked buffers of a batched model whose experts are factorised together: chol / invd / alpha / info (and minv, work
        when the inverse is eager) as [nexp, ...] tensors, the experts holding views.  None when the model is not batched that way
        (one expert, a Compose longer than one pg_covspec, or experts above _BATCH_MAX_N)."""
        experts = self._experts
        n_pad = experts[0].n_pad
        spec, _ = spec_of(self.cov, self._x.shape[-1])
        if len(experts) < 2 or n_pad > _BATCH_MAX_N or isinstance(spec, (list, tuple)) and len(spec) != 1:
            return None
        if self._bat is None:
            ops = get_ops()
            nb = len(experts)
            bat = {
                "chol": ops.empty(nb, n_pad, n_pad, dtype=self.dtype),
                "invd": ops.empty(nb, ops.potrf_worksize(n_pad, self.dtype), dtype=self.dtype),
                "alpha": ops.empty(nb, n_pad, dtype=self.dtype),
                "info": torch.zeros(nb, dtype=torch.int32, device=ops.device),
                "hp": ops.empty(nb, self.params.shape[-1], dtype=torch.float64),
                "minv": None, "work": None,
            }
            for b, e in enumerate(experts):
                e.chol, e.invd, e.alpha, e.info, e.hp = bat["chol"][b], bat["invd"][b], bat["alpha"][b], bat["info"][b: b + 1], bat["hp"][b]
            self._bat = bat
        self._bat["eager"] = self.eager_inverse or n_pad <= _BATCH_EAGER_N
        if self._bat["eager"] and self._bat["minv"] is None:
            ops = get_ops()
            nb = len(experts)
            self._bat["minv"] = ops.empty(nb, n_pad, n_pad, dtype=self.dtype)
            self._bat["u"] = ops.empty(nb, n_pad, dtype=self.dtype)
            self._bat["work"] = ops.empty(nb, (n_pad // 256) * n_pad, dtype=self.dtype)
            for b, e in enumerate(experts):
                e.minv = self._bat["minv"][b]
        return self._bat

    def _hp_rows(self):
        nhp = self.params.shape[-1]
        return self.params.reshape(-1, nhp).to(torch.float64)

    # ---- the path ---------------------------------------------------------------------------
    def update(self) -> None:
        experts = self._device_experts()      # first: an in-place edit of x / y (version bump) marks the model dirty here
        if self.need_upd:
            ops = get_ops()
            hp_rows = self._hp_rows()
            spec, nhp = spec_of(self.cov, self._x.shape[-1])
            assert hp_rows.shape[-1] == nhp

            bat = self._batch()

            def enqueue_batched():
                # all experts in one call: every launch of the blocked factorisation (and of L^-1) covers the whole batch
                bat["hp"].copy_(hp_rows.expand(len(experts), -1) if hp_rows.shape[0] == 1 else hp_rows)
                x_stride = self._x_all.stride(0) if self._x_all.shape[0] > 1 else 0
                ops.build_factor_batched(spec, bat["hp"], self._x_all, x_stride, bat["chol"], bat["invd"], bat["info"],
                                         bat["minv"] if bat["eager"] else None)
                if bat["eager"]:
                    ops.alpha_batched(bat["minv"], self._y_all, bat["u"], bat["alpha"], bat["work"])
                for e in experts:
                    e.minv_valid = bat["eager"]
                    if not bat["eager"]:
                        ops.potrs_vec(e.chol, e.invd, e.y, e.alpha)

            def enqueue_serial():
                for b, e in enumerate(experts):
                    e.hp = ops.to_device(hp_rows[b % hp_rows.shape[0]], torch.float64)
                    if e.chol is None:
                        e.chol = ops.empty(e.n_pad, e.n_pad, dtype=self.dtype)
                        e.invd = ops.potrf_workspace(e.n_pad, self.dtype)
                        e.alpha = ops.empty(e.n_pad, dtype=self.dtype)
                        e.info = torch.zeros(1, dtype=torch.int32, device=ops.device)
                    if self.eager_inverse:
                        if e.minv is None:
                            e.minv = ops.empty(e.n_pad, e.n_pad, dtype=self.dtype)
                            e.work = ops.empty((e.n_pad // 256 + 1) * e.n_pad, dtype=self.dtype)
                        ops.build_factor(spec, e.hp, e.x, e.chol, e.invd, e.info, e.minv)
                        u = e.work[: e.n_pad]
                        ops.trmv(e.minv, e.y, u, 0)
                        ops.trmv(e.minv, u, e.alpha, 1, e.work[e.n_pad:])
                        e.minv_valid = True
                    else:
                        e.minv_valid = False
                        ops.build_factor(spec, e.hp, e.x, e.chol, e.invd, e.info)
                        ops.potrs_vec(e.chol, e.invd, e.y, e.alpha)

            enqueue = enqueue_batched if bat is not None else enqueue_serial

            # one sync point after everything is enqueued; a timed-out coupled chain repeats the lot on the classic chain
            for info in _checked(enqueue, lambda: torch.cat([e.info for e in experts]).tolist()):
                if info:
                    raise _lin_alg_error(info)
            self.need_upd = False
        return None

    def _minv(self, e):
        if not e.minv_valid:
            ops = get_ops()
            if e.minv is None:
                e.minv = ops.empty(e.n_pad, e.n_pad, dtype=self.dtype)
            ops.trtri(e.chol, e.invd, e.minv)
            e.minv_valid = True
        return e.minv

    def _kss_diag(self, b: int) -> float:
        """diag of cov.kernel(params, xp): sum sigma_c^2 + sum sigma_n^2 (White_noise sees xp=None,
        gpr.py:98), no jitter."""
        _, offs, noise, _ = layout(self.cov, self._x.shape[-1])
        hp = self._hp_rows()
        row = hp[b % hp.shape[0]]
        return float(sum(row[o] ** 2 for o in offs) + sum(row[o] ** 2 for o in noise))

    def _predict_expert(self, b, e, xpd, want):
        ops = get_ops()
        spec, _ = spec_of(self.cov, self._x.shape[-1])
        m = xpd.shape[0]
        mean = ops.empty(m, dtype=self.dtype)
        var = ops.empty(m, dtype=self.dtype) if want == "diag" else None
        for s in range(0, m, _CHUNK):
            xq = xpd[s: s + _CHUNK]
            mc = xq.shape[0]
            m_pad = pad_to(mc)
            # K* (test-point-major: k(xp, x); the kernels are symmetric) and the product's scratch are kept per model and shape: all
            # experts of a model and all chunks of a call run one after the other on the stream, so ONE set serves them (round 2
            # allocated 604 MB per expert per batch at config 4 through torch's caching allocator)
            key = (m_pad, e.n_pad, want == "diag")
            if self._pbuf is None or self._pbuf[0] != key:
                self._pbuf = (key, ops.empty(m_pad, e.n_pad, dtype=self.dtype), ops.empty((e.n_pad // 64) * m_pad, dtype=self.dtype))
            kt, work = self._pbuf[1], self._pbuf[2]
            ops.kernel_build(spec, e.hp, xq, e.x, kt)
            mu = ops.empty(m_pad, dtype=self.dtype)
            vq = ops.empty(m_pad, dtype=self.dtype) if want == "diag" else None
            ops.predict_mean_q_kt(kt, self._minv(e) if want == "diag" else None, e.alpha, mu, vq,
                                  self._kss_diag(b), work)
            mean[s: s + mc] = mu[:mc]
            if want == "diag":
                var[s: s + mc] = vq[:mc]
        if want == "diag":
            return mean, var
        return mean, None

    def _predict_full(self, xqs):
        """Mean K* alpha and K** - K* K^-1 K*^T = K** - V^T V with V = L^-1 K*^T (gpr.py:80-85,108-120) for every expert (expert b at the
        points xqs[b]): per expert the test-point-major K* -- built ONCE, the mean is taken from it too (round 4 built it a second time
        for the mean) --, Vt = K* L^-T (both operands read along k) and K**; then ONE rank-n update for all experts -- the 136 lower tiles
        of one 2048 x 2048 output leave three quarters of the chip idle, eight experts' tiles fill it.  Returns (means, covariances)."""
        ops = get_ops()
        spec, _ = spec_of(self.cov, self._x.shape[-1])
        experts = self._experts
        m = xqs[0].shape[0]
        m_pad = pad_to(m)
        item = torch.empty(0, dtype=self.dtype).element_size()
        c_all = ops.empty(len(experts), m_pad, m_pad, dtype=self.dtype)
        mean_all = ops.empty(len(experts), m_pad, dtype=self.dtype)
        dummy = ops.empty(256, dtype=self.dtype)
        budget = _group_budget(_FULL_VT_BYTES, 2 * sum(m_pad * e.n_pad for e in experts) * item)      # per group: Vt and (stacked inverses) K* of every expert in it
        b = 0
        while b < len(experts):
            n_pad = experts[b].n_pad
            # experts of one padded size share a launch, within the budget (never fewer than one)
            cnt = 1
            while (b + cnt < len(experts) and experts[b + cnt].n_pad == n_pad and 2 * (cnt + 1) * m_pad * n_pad * item <= budget):
                cnt += 1
            vt = ops.empty(cnt, m_pad, n_pad, dtype=self.dtype)
            minvs = [self._minv(experts[b + i]) for i in range(cnt)]
            step = (minvs[1].data_ptr() - minvs[0].data_ptr()) if cnt > 1 else 0
            stacked = cnt > 1 and step > 0 and all(mi.data_ptr() - minvs[0].data_ptr() == i * step and mi.stride(0) == minvs[0].stride(0)
                                                   for i, mi in enumerate(minvs))
            kt = ops.empty(cnt if stacked else 1, m_pad, n_pad, dtype=self.dtype)
            for i in range(cnt):
                e = experts[b + i]
                ops.kernel_build(spec, e.hp, xqs[b + i], e.x, kt[i if stacked else 0])
                ops.predict_mean_q_kt(kt[i if stacked else 0], None, e.alpha, mean_all[b + i], None, 0.0, dummy)   # mean = K* alpha
                if not stacked:
                    ops.trmm_lower_kt(minvs[i], kt[0], vt[i])
                ops.kernel_build(spec, e.hp, xqs[b + i], None, c_all[b + i])   # K** incl. sigma_n^2, padding = identity
            if stacked:      # experts factorised together hold their inverses in one stack: one launch for all the products
                ops.trmm_lower_kt(minvs, kt, vt)
            ops.syrk_nt_sub_batched(vt, c_all[b: b + cnt], lower_only=True)     # n m^2 flop per expert, lower tiles
            b += cnt
        out = []
        for i in range(len(experts)):
            ops.symmetrize(c_all[i], m_pad)                 # the upper triangle is the mirror: exactly symmetric
            out.append(c_all[i][:m, :m])
        return [mean_all[i, :m] for i in range(len(experts))], out

    def _predict_batched(self, xpd, want):
        """Mean (and diagonal variance) of ALL experts of a model whose experts live in stacked buffers (`_batch`): per chunk of test
        points ONE launch builds every expert's test-point-major K*, three more give every mean and variance (round 5; the reference:
        one batched kernel / bmm / cholesky_solve, gpr.py:76-106).  Per expert the numbers are those of `_predict_expert`, bit for bit."""
        ops = get_ops()
        spec, _ = spec_of(self.cov, self._x.shape[-1])
        bat, experts = self._bat, self._experts
        nb, n_pad = len(experts), experts[0].n_pad
        m = xpd.shape[-2]
        diag = want == "diag"
        if diag and bat["minv"] is None:      # inverses not formed with the factor (lazy model): form them into one stack now
            bat["minv"] = ops.empty(nb, n_pad, n_pad, dtype=self.dtype)
            for b, e in enumerate(experts):
                e.minv, e.minv_valid = bat["minv"][b], False
        if diag:
            for e in experts:
                self._minv(e)
        # outputs: one fresh buffer per call, rows long enough for the padded last chunk -- the kernels write every chunk's means and
        # variances straight into their place (no staging copies: at the reference's test sizes those were half the call)
        mtot = ((m // _CHUNK) * _CHUNK + pad_to(m % _CHUNK)) if m % _CHUNK else m
        # (means and variances as the two halves of ONE buffer: `predict` then brings both to the host in one transfer and one
        # synchronisation -- at the reference's test sizes a device-to-host copy is a tenth of the call)
        out_all = ops.empty(2 if diag else 1, nb, mtot, dtype=self.dtype)
        mean_all = out_all[0]
        var_all = out_all[1] if diag else None
        item = torch.empty(0, dtype=self.dtype).element_size()
        x_all = self._x_all
        for s in range(0, m, _CHUNK):
            xq = xpd[..., s: s + _CHUNK, :]
            xq = xq if xq.is_contiguous() else xq.contiguous()
            mc = xq.shape[-2]
            m_pad = pad_to(mc)
            # experts per launch group: every expert's K* (m_pad x n_pad) at once, within the memory budget
            per = m_pad * n_pad * item
            grp = max(1, min(nb, _group_budget(64 << 30, nb * per) // per))
            key = ("bat", grp, m_pad, n_pad, diag)
            if self._pbuf is None or self._pbuf[0] != key:
                self._pbuf = None
                self._pbuf = (key, ops.empty(grp, m_pad, n_pad, dtype=self.dtype), ops.empty(grp, (n_pad // 64) * m_pad, dtype=self.dtype))
            _, kt, work = self._pbuf
            for b0 in range(0, nb, grp):
                cnt = min(grp, nb - b0)
                xr = xq if xq.dim() == 2 else (xq[b0: b0 + cnt] if xq.shape[0] > 1 else xq[0])
                xc = x_all[b0: b0 + cnt] if x_all.shape[0] > 1 else x_all
                ops.kernel_build_batched(spec, bat["hp"][b0: b0 + cnt], xr, xc, kt[:cnt])
                ops.predict_mean_q_kt_batched(kt[:cnt], bat["minv"][b0: b0 + cnt] if diag else None, bat["alpha"][b0: b0 + cnt],
                                              mean_all[b0: b0 + cnt, s: s + m_pad], var_all[b0: b0 + cnt, s: s + m_pad] if diag else None,
                                              spec, bat["hp"][b0: b0 + cnt], work[:cnt])
        return [mean_all[b, :m] for b in range(nb)], ([var_all[b, :m] for b in range(nb)] if diag else [None] * nb)

    def _predict_device(self, xpd, want):
        """Per-expert device tensors (mean[m], var[m] | cov[m,m] | None) for device-resident test points: xpd [m, d]
        (the same points for every expert) or [nc, m, d] (expert b predicts at xpd[b], as cov.kernel(params, x, xp)
        broadcasts in the reference, gpr.py:79)."""
        self._device_experts()
        self.update()
        if xpd.dim() == 3 and xpd.shape[0] not in (1, len(self._experts)):
            raise RuntimeError("batch dimension of xp (%d) does not match the %d experts" % (xpd.shape[0], len(self._experts)))
        if want == "full":       # K* is built once per expert there: mean, Vt and the update all come from it
            return self._predict_full([xpd if xpd.dim() == 2 else xpd[b % xpd.shape[0]] for b in range(len(self._experts))])
        if self._bat is not None and len(self._experts) > 1 and not os.environ.get("PG_PREDICT_SERIAL"):
            self.last_predict_batched = True
            return self._predict_batched(xpd, want)
        self.last_predict_batched = False
        means, covs = [], []
        for b, e in enumerate(self._experts):
            xq = xpd if xpd.dim() == 2 else xpd[b % xpd.shape[0]]
            mu, cv = self._predict_expert(b, e, xq, want)
            means.append(mu)
            covs.append(cv)
        return means, covs

    def predict(self, xp: Tensor, var: str = "full") -> Sequence[Tensor]:
        ops = get_ops()
        want = var if var in ("full", "diag") else "none"
        xpd = ops.to_device(xp if xp.dim() == 2 else xp.reshape(-1, xp.shape[-2], xp.shape[-1]), self.dtype)
        means, covs = self._predict_device(xpd, want)
        # (the batched prediction hands out rows of ONE fresh buffer: the stacked result is a view of it, no stacking kernel)
        mstack = _stacked_rows(means) if len(means) > 1 else None
        if want == "diag" and self.batched and mstack is not None:
            cstack = _stacked_rows(covs)
            both = _stacked_pair(mstack, cstack) if cstack is not None else None
            if both is not None:       # [2, nc, m] view of one buffer: one transfer
                host = both.to(xp.device)
                return [host[0].squeeze(), host[1]]
        ys = (mstack if mstack is not None else torch.stack(means)).squeeze().to(xp.device)   # squeeze_(): drops every size-1 dim (gpr.py:87)
        if want == "none":
            covars = NotImplemented
        elif self.batched:
            cstack = _stacked_rows(covs) if want == "diag" else None
            covars = (cstack if cstack is not None else torch.stack(covs)).to(xp.device)
        else:
            covars = covs[0].contiguous().to(xp.device)
        return [ys, covars]

    def predict_var(self, xp: Tensor, **kwargs: Tensor) -> Tensor:
        return self.predict(xp, var="diag")[1]

    def predict_covar(self, xp: Tensor, **kwargs: Tensor) -> Tensor:
        return self.predict(xp, var="full")[1]

    # ---- reference attributes, materialised on access ------------------------------------------
    def _stack(self, ts):
        out = torch.stack(ts) if self.batched else ts[0]
        return out.contiguous().to(self._x.device)

    @property
    def krn(self) -> Tensor:
        """K + 1e-7 I as `Exact_GP.krn` holds it after update (gpr.py:67-68); rebuilt on access."""
        self.update()
        ops = get_ops()
        spec, _ = spec_of(self.cov, self._x.shape[-1])
        outs = []
        for e in self._experts:
            k = ops.empty(e.n_pad, e.n_pad, dtype=self.dtype)
            ops.kernel_build(spec, e.hp, e.x, None, k, jitter=JITTER)
            outs.append(k[: e.n, : e.n])
        return self._stack(outs)

    @property
    def krnchd(self) -> Tensor:
        """Lower Cholesky factor with a zero upper triangle, like tc.cholesky (gpr.py:69)."""
        self.update()
        ops = get_ops()
        outs = []
        for e in self._experts:
            c = e.chol.clone()
            ops.tril(c, e.n_pad)
            outs.append(c[: e.n, : e.n])
        return self._stack(outs)

    @property
    def wt(self) -> Tensor:
        """alpha = K^-1 y (gpr.py:70-72): cholesky_solve(y[..., None], L) keeps y's leading dimensions."""
        self.update()
        out = self._stack([e.alpha[: e.n] for e in self._experts])
        return out.reshape(self._y.shape) if (not self.batched and self._y.dim() > 1) else out
