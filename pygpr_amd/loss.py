"""Model-selection loss with PyGPR's surface (reference: PyGPR/loss.py) on the MI355X.

    NLML          = 1/2 y^T a + sum_i log L_ii + n/2 log 2pi          (loss.py:47-49, 107-109)
    dNLML/dtheta_k = -1/2 (a^T dK_k a - tr(K^-1 dK_k))                 (loss.py:116-121)

The reference materialises dK[nhp,n,n] and runs `cholesky_solve` on it (n^3/3 + 2 nhp n^3 flop).  Here
one evaluation is: covariance build (lower tiles) -> blocked Cholesky -> L^-1 -> alpha = L^-T (L^-1 y)
-> K^-1 = L^-T L^-1 -> fused contraction 1/2 sum (K^-1 - a a^T) o dK_k with dK recomputed from the
point tiles -- n^3 flop in MFMA GEMMs, two n x n device buffers, one [1 + nhp] transfer back.
`MLE.*` take and return host NumPy fp64 exactly like the reference (scipy drives them).
"""
import os
from typing import Tuple

import numpy as np
import torch
from numpy import ndarray

from ._ops import get_ops
from .covar import layout, spec_of
from .gpr import _BATCH_EAGER_N, _BATCH_MAX_N, GPR, _checked, _lin_alg_error


class Loss():
    """Base class for cost functions for GP model selection (PyGPR/loss.py:10-28)."""

    def __init__(self, model: GPR) -> None:
        self.model: GPR = model
        self.loss_value: float = NotImplemented
        self.grad_value: ndarray = NotImplemented
        return None

    def loss(self, params: ndarray) -> float:
        raise NotImplementedError

    def grad(self, params: ndarray) -> ndarray:
        raise NotImplementedError

    def loss_and_grad(self, params: ndarray) -> Tuple[float, ndarray]:
        raise NotImplementedError


class MLE(Loss):
    """Negative log marginal likelihood of the hyper-parameters (PyGPR/loss.py:31-128)."""

    def __init__(self, model: GPR) -> None:
        super().__init__(model)
        self._buf = {}
        self._bbuf = {}
        self.last_batched = False   # the last device evaluation took the experts-together path (tests / diagnostics)
        self.memoize = True     # re-use the last evaluation when asked again at identical parameters (off in benchmarks)
        self._memo = None
        self._factor_key = None   # memo key of the loss-only evaluation whose factor is still in the work buffers

    def _buffers(self, n_pad, dtype, nhp, n):
        key = (n_pad, dtype, nhp, n)
        if self._buf.get("key") != key:
            ops = get_ops()
            self._buf = {
                "key": key,
                "a": ops.empty(n_pad, n_pad, dtype=dtype),       # K -> L -> K^-1
                "m": None,                                         # L^-1 (gradient evaluations only)
                "invd": ops.potrf_workspace(n_pad, dtype),
                "info": torch.zeros(1, dtype=torch.int32, device=ops.device),
                "alpha": ops.empty(n_pad, dtype=dtype),
                "u": ops.empty(n_pad, dtype=dtype),
                "vwork": ops.empty((n_pad // 256) * n_pad, dtype=dtype),
                "gwork": ops.empty(ops.nlml_grad_worksize(n, nhp), dtype=torch.float64),
                "out": ops.zeros(1 + nhp, dtype=torch.float64),
                "val": ops.zeros(2, dtype=torch.float64),
            }
        return self._buf

    def _evaluate(self, params: ndarray, want_grad: bool):
        """One evaluation, memoised on (parameters, data identity and version, covariance): the reference re-factorises on
        every call (loss.py:39,64,97); CG_Quad / BFGS_Quad / hessian ask for grad(par) at the same point again and again,
        and get_learn_rate / Nelder_Mead callers ask for loss(p) and then grad(p).  Two levels:
          * the last RESULT is returned as is when it already holds what is asked for;
          * after a loss-only evaluation of a single model the FACTOR stays in the work buffers: a following grad(p) at
            the same p only adds L^-1, K^-1 and the contraction (no second covariance build / Cholesky)."""
        m = self.model      # the reference re-reads model.x / .y / .cov on every call: all three are part of the key
        key = (np.asarray(params, dtype=np.float64).tobytes(), np.shape(params), id(m._x), m._x._version, id(m._y), m._y._version,
               id(m.cov), tuple(map(tuple, layout(m.cov, m._x.shape[-1])[:3])))
        hit = self._memo if (self.memoize and self._memo is not None and self._memo[0] == key) else None
        if hit is not None and (hit[2] is not None or not want_grad):
            return hit[1].copy(), (hit[2].copy() if hit[2] is not None else None)
        reuse = hit is not None and want_grad and self._factor_key == key
        loss, grad = self._evaluate_device(params, want_grad, key if self.memoize else None, reuse)
        self._memo = (key, loss.copy(), grad.copy() if want_grad else None)
        return loss, grad

    def _evaluate_device(self, params: ndarray, want_grad: bool, key=None, reuse_factor=False):
        ops = get_ops()
        model = self.model
        d = model.x.shape[-1]
        spec, nhp = spec_of(model.cov, d)
        p = np.asarray(params, dtype=np.float64)
        assert p.shape[-1] == nhp  # covar.py:52,66
        rows = p.reshape(-1, nhp)
        experts = model._device_data()      # one (x, y) per leading index of the data; the batch of params broadcasts
        if len(experts) not in (1, rows.shape[0]) and rows.shape[0] != 1:
            raise RuntimeError("batch dimensions of params and x do not broadcast")
        nb = max(len(experts), rows.shape[0])
        self._factor_key = None             # whatever the buffers held is overwritten below
        # Every expert's evaluation is enqueued before the ONE synchronisation that reads all results: the experts share the n x n
        # work buffers (stream order makes that safe), each has its own result / status slot.  (Round 2 synchronised per expert.)
        hp_all = ops.to_device(torch.from_numpy(np.array(np.broadcast_to(rows, (nb, nhp)), dtype=np.float64)), torch.float64)
        outs = ops.zeros(nb, 1 + nhp, dtype=torch.float64)
        vals = ops.zeros(nb, 2, dtype=torch.float64)
        infos = torch.zeros(nb, dtype=torch.int32, device=ops.device)
        res = [None]

        # Experts together (round 4): the reference evaluates a batched model's NLML and gradient as ONE batched factorisation and
        # solve (loss.py:92-128 on x [nc, n, d] of gr_bcm.py:19-29).  Up to _BATCH_MAX_N points per expert every step below is one
        # call whose launches cover all experts: build + Cholesky + L^-1, alpha + NLML, K^-1 = L^-T L^-1, the contraction -- about
        # twenty launches and one synchronisation for the whole batch, where the loop below pays each expert's latency-bound chain
        # in turn (n = 4096: 0.21 of the matrix peak per expert).  Per expert the numbers are those of the loop on the classic chain.
        n_pad = experts[0].n_pad
        self.last_batched = False
        one_pass = not isinstance(spec, (list, tuple)) or len(spec) == 1
        batched_path = (nb > 1 and not reuse_factor and one_pass and n_pad <= _BATCH_MAX_N
                        and (want_grad or n_pad <= _BATCH_EAGER_N) and not os.environ.get("PG_MLE_SERIAL"))

        def enqueue_batched():
            n = experts[0].n
            key = ("bat", nb, n_pad, model.dtype, nhp, n)
            if self._bbuf.get("key") != key:
                self._bbuf = {
                    "key": key,
                    "a": ops.empty(nb, n_pad, n_pad, dtype=model.dtype),       # K -> L -> K^-1, per expert
                    "m": ops.empty(nb, n_pad, n_pad, dtype=model.dtype),       # L^-1
                    "invd": ops.empty(nb, ops.potrf_worksize(n_pad, model.dtype), dtype=model.dtype),
                    "alpha": ops.empty(nb, n_pad, dtype=model.dtype),
                    "u": ops.empty(nb, n_pad, dtype=model.dtype),
                    "vwork": ops.empty(nb, (n_pad // 256) * n_pad, dtype=model.dtype),
                    "gwork": ops.empty(nb * ops.nlml_grad_worksize(n, nhp), dtype=torch.float64) if want_grad else None,
                }
            b = self._bbuf
            if want_grad and b["gwork"] is None:
                b["gwork"] = ops.empty(nb * ops.nlml_grad_worksize(n, nhp), dtype=torch.float64)
            x_all, y_all = model._x_all, model._y_all
            x_stride = x_all.stride(0) if x_all.shape[0] > 1 else 0
            ops.build_factor_batched(spec, hp_all, x_all, x_stride, b["a"], b["invd"], infos, b["m"])
            ops.alpha_nlml_batched(b["m"], y_all, b["u"], b["alpha"], b["vwork"], n, outs)
            if want_grad:
                ops.lauum_batched(b["m"], b["a"])
                ops.nlml_grad_batched(spec, hp_all, x_all, x_stride, n, b["a"], b["alpha"], outs[:, 1:], b["gwork"])
            self.last_batched = True
            res[0] = outs.cpu().numpy()                                     # the one sync + transfer

        def enqueue():
            for b in range(nb):
                e = experts[b % len(experts)]
                buf = self._buffers(e.n_pad, model.dtype, nhp, e.n)
                hp, out, info, val = hp_all[b], outs[b], infos[b: b + 1], vals[b]
                a = buf["a"]
                if want_grad and buf["m"] is None:
                    buf["m"] = ops.empty(e.n_pad, e.n_pad, dtype=model.dtype)
                m = buf["m"]
                if reuse_factor and nb == 1:
                    # `a` still holds the factor and buf["alpha"] the weights of the loss-only evaluation at these parameters
                    ops.trtri(a, buf["invd"], m)
                    ops.nlml_value(a, e.y, buf["alpha"], e.n, out)
                    ops.lauum(m, a)
                    ops.nlml_grad(spec, hp, e.x, e.n, a, buf["alpha"], out[1:], buf["gwork"])
                elif want_grad:
                    ops.build_factor(spec, hp, e.x, a, buf["invd"], info, m)   # covariance build + Cholesky + L^-1 in one call
                    # alpha = L^-T (L^-1 y) and the NLML value on the library's side stream, beside K^-1 = L^-T L^-1 (lower, over a)
                    ops.alpha_nlml_async(a, m, e.y, buf["u"], buf["alpha"], buf["vwork"], e.n, val)
                    ops.lauum(m, a)
                    ops.nlml_grad(spec, hp, e.x, e.n, a, buf["alpha"], out[1:], buf["gwork"])   # waits for the side stream
                    out[0:1].copy_(val[0:1])
                else:
                    ops.build_factor(spec, hp, e.x, a, buf["invd"], info)
                    ops.potrs_vec(a, buf["invd"], e.y, buf["alpha"])
                    ops.nlml_value(a, e.y, buf["alpha"], e.n, out)
            res[0] = outs.cpu().numpy()                                     # the one sync + transfer

        # (a timed-out coupled chain -- info = -1 -- repeats the evaluation on the classic chain; the re-used factor of a
        # loss-only evaluation was checked when it was made, so that branch cannot time out)
        for info in _checked(enqueue_batched if batched_path else enqueue, lambda: infos.tolist()):
            if info:
                raise _lin_alg_error(info)
        losses = res[0][:, 0].copy()
        grads = res[0][:, 1:].copy()
        if not want_grad and nb == 1 and key is not None:
            self._factor_key = key
        batched = nb > 1      # a batch of one is squeezed away (llhd.squeeze_(0), loss.py:51,85,111)
        loss = losses.copy() if batched else np.array(losses[0])
        grad = grads.copy() if batched else grads[0].copy()
        return loss, grad

    def loss(self, params: ndarray) -> float:
        llhd, _ = self._evaluate(params, False)
        self.loss_value = llhd
        return llhd

    def grad(self, params: ndarray) -> ndarray:
        _, jac = self._evaluate(params, True)
        self.grad_value = jac
        return jac

    def loss_and_grad(self, params: ndarray) -> Tuple[float, ndarray]:
        llhd, jac = self._evaluate(params, True)
        self.loss_value = llhd
        self.grad_value = jac
        return (llhd, jac)
