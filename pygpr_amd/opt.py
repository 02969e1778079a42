"""Optimisers with PyGPR's surface (reference: PyGPR/opt.py).  Host code: the vectors here have nhp
entries; every loss / gradient they ask for is one device evaluation in loss.py.

  CG, Nelder_Mead   : scipy.optimize.minimize drivers with the reference's options, opt.dat trace and
                      write-back rules (opt.py:29-122)
  hessian           : forward-difference Hessian from a gradient callable (opt.py:125-137)
  CG_Quad, BFGS_Quad: quadratic-model optimisers on finite-difference Hessian-vector products of
                      loss.grad (opt.py:140-295)
"""
from typing import Callable

import numpy as np
import scipy.optimize as scopt
import torch
from numpy import ndarray

from .loss import Loss


class Opt:
    """Base class for optimisers (opt.py:11-26)."""

    def __init__(self, loss: Loss, par: ndarray = None) -> None:
        self.loss: Loss = loss
        self.args: dict = {}
        self.x: ndarray = NotImplemented
        return None

    def minimize(self):
        raise NotImplementedError

    def step(self):
        raise NotImplementedError


class _ScipyDriver(Opt):
    _method = None
    _use_jac = False
    _write_back_on_failure = False

    def _trace(self, params):
        return [*params, self.loss.loss_value]

    def callback(self, params: ndarray) -> None:
        print(*self._trace(params), file=self.fstr)

    def minimize(self) -> None:
        start = torch.clone(self.loss.model.params).numpy()
        self.fstr = open("opt.dat", "w")
        try:
            self.res = scopt.minimize(
                self.loss.loss_and_grad if self._use_jac else self.loss.loss,
                start,
                method=self._method,
                jac=True if self._use_jac else None,
                callback=self.callback,
                options=self.args,
            )
        finally:
            self.fstr.close()
        if self.res.success is True or self._write_back_on_failure:
            self.loss.model.set_params(torch.from_numpy(self.res.x))
        if self.res.success is not True:
            print("Optimizer Failed")
        return None

    def step(self):
        raise NotImplementedError


class CG(_ScipyDriver):
    """Conjugate gradient on loss.loss_and_grad (opt.py:29-78); res.x is written back to the model
    even when scipy reports failure (opt.py:61-65)."""

    _method = "CG"
    _use_jac = True
    _write_back_on_failure = True

    def __init__(self, loss: Loss) -> None:
        super().__init__(loss)
        self.args = {"gtol": 1e-4, "maxiter": 1000, "disp": True, "return_all": True}
        self.res: scopt.OptimizeResult = NotImplemented

    def _trace(self, params):
        return [*params, self.loss.loss_value, np.linalg.norm(self.loss.grad_value)]


class Nelder_Mead(_ScipyDriver):
    """Nelder-Mead on loss.loss (opt.py:81-122); writes back only on success (opt.py:111-114)."""

    _method = "Nelder-Mead"

    def __init__(self, loss: Loss) -> None:
        super().__init__(loss)
        self.args = {"fatol": 1e-4, "maxiter": 1000, "disp": True, "return_all": True}
        self.res: scopt.OptimizeResult = NotImplemented


def hessian(x: np.ndarray, jac: Callable[..., np.ndarray], eps: float) -> np.ndarray:
    """Column i = (jac(x + eps e_i) - jac(x)) / eps (opt.py:125-137)."""
    dim = x.shape[-1]
    j0 = jac(x)
    hess = np.empty([dim, dim])
    for i in range(dim):
        xe = np.copy(x)
        xe[i] += eps
        hess[:, i] = (jac(xe) - j0) / eps
    return hess


class _QuadBase(Opt):
    def __init__(self, loss: Loss, gtol: float = 1e-4, max_iter: int = 100, fd_eps: float = 1e-5) -> None:
        super().__init__(loss)
        self.x: ndarray = NotImplemented
        self.r: ndarray = NotImplemented
        self.eps = fd_eps
        self.max_iter = max_iter
        self.gtol = gtol

    def _start(self, par):
        return self.loss.model.params.numpy() if par is None else par

    def _run(self) -> int:
        k = 0
        gnorm = np.linalg.norm(self.r)
        with open("opt.dat", "w") as fstr:
            while gnorm > self.gtol and k < self.max_iter:
                self.step()
                gnorm = np.linalg.norm(self.r)
                k += 1
                print(k, gnorm, file=fstr)
        if self.loss.model is not None:
            self.loss.model.set_params(torch.tensor(self.x))
        return k


class CG_Quad(_QuadBase):
    """Linear conjugate gradient on the local quadratic model (opt.py:140-214)."""

    def __init__(self, loss: Loss, gtol: float = 1e-4, max_iter: int = 100, fd_eps: float = 1e-5) -> None:
        super().__init__(loss, gtol, max_iter, fd_eps)
        self.p: ndarray = NotImplemented

    def hessian_product(self, par: ndarray, v: ndarray, eps: float) -> ndarray:
        return (self.loss.grad(par + eps * v) - self.loss.grad(par)) / eps

    def step(self) -> None:
        hp = self.hessian_product(self.x, self.p, eps=self.eps)
        rr = np.dot(self.r, self.r)
        alp = rr / np.dot(self.p, hp)
        self.x = self.x + alp * self.p
        self.r = self.r + alp * hp
        self.p = (np.dot(self.r, self.r) / rr) * self.p - self.r
        return None

    def minimize(self, par: ndarray = None) -> int:
        self.x = self._start(par)
        self.r = self.loss.grad(self.x)
        self.p = -1.0 * self.r
        return self._run()


class BFGS_Quad(_QuadBase):
    """BFGS with unit steps for a (near) quadratic function (opt.py:217-295)."""

    def __init__(self, loss: Loss, gtol: float = 1e-4, max_iter: int = 100, fd_eps: float = 1e-5) -> None:
        super().__init__(loss, gtol, max_iter, fd_eps)
        self.HI: ndarray = NotImplemented

    def hessian_inv_update(self, HI: ndarray, s: ndarray, y: ndarray) -> ndarray:
        eye = np.identity(HI.shape[-1])
        rho = 1 / np.dot(y, s)
        left = eye - rho * np.outer(s, y)
        return left @ HI @ left.T + rho * np.outer(s, s)

    def step(self):
        x_new = self.x - self.HI @ self.r
        r_new = self.loss.grad(x_new)
        self.HI = self.hessian_inv_update(self.HI, x_new - self.x, r_new - self.r)
        self.x, self.r = x_new, r_new
        return None

    def minimize(self, par: ndarray = None, H0: ndarray = None):
        self.x = self._start(par)
        self.r = self.loss.grad(self.x)
        self.HI = np.identity(self.x.shape[-1]) if H0 is None else np.linalg.inv(H0)
        return self._run()
