"""Covariance kernels with PyGPR's `Covar` protocol (reference: PyGPR/covar.py), evaluated on the
MI355X through libpygpr_hip.

Same names, hyper-parameter layout and shapes as the reference:
  * `Squared_exponential`  hp = [sigma, l_1..l_d], K = sigma^2 exp(-sum_d l_d^2 (x_d - x'_d)^2),
    l are INVERSE length scales, sigma enters squared, no 1/2 in the exponent (covar.py:129-167)
  * `White_noise`          hp = [sigma_n], sigma_n^2 I; with `xp` given it is `tensor(0)` (covar.py:227-245)
  * `Compose`              sum of children, hp concatenated in list order, dK concatenated on dim -3
                           (covar.py:28-81)
  * `Matern52`             NEW (no reference counterpart), same hp layout as the SE.
Leading batch dims on hp and/or x follow the reference's flatten-to-one-batch-dim rule.  Tensors come
back on the device of `x` (CPU in -> CPU out); the arithmetic always runs on the GPU in the dtype of
`x` (float64, or float32 as an explicit opt-in) -- unlike the reference, nothing here touches torch's
global default dtype.  Distances are direct sums of squared differences instead of the reference's GEMM
expansion (covar.py:102-127): same value to rounding, exactly symmetric, never negative.
"""
from typing import List, Protocol, Sequence

import torch
from torch import Tensor

from . import _lib
from ._ops import get_ops, make_specs, pad_to


class Covar(Protocol):
    """Protocol for covariance kernels (PyGPR/covar.py:9-25)."""

    def get_params_shape(self, x: Tensor) -> List[int]:
        ...

    def init_params(self, x: Tensor) -> Tensor:
        ...

    def kernel(self, params: Tensor, x: Tensor, xp: Tensor = None) -> Tensor:
        ...

    def kernel_and_grad(self, params: Tensor, x: Tensor) -> List[Tensor]:
        ...


def _params_shape(x: Tensor, nhp: int) -> List[int]:
    shape = list(x.shape)
    shape[-1] = nhp
    shape.pop(-2)
    return shape


def layout(cov, d):
    """Flatten a covariance object into (kinds, offsets, noise_offsets, nhp) for pg_covspec."""
    kinds, offs, noise = [], [], []
    nhp = cov._collect(d, 0, kinds, offs, noise)
    return kinds, offs, noise, nhp


def spec_of(cov, d):
    """(passes, nhp): the pg_covspec list the device ops take (one entry unless the Compose has more than
    PG_MAX_COMP stationary or noise children)."""
    kinds, offs, noise, nhp = layout(cov, d)
    return make_specs(kinds, offs, noise), nhp


class _DeviceKernel:
    """Shared evaluation code: every concrete kernel only says how it lays out its parameters."""

    def _collect(self, d, base, kinds, offs, noise):  # -> number of parameters consumed
        raise NotImplementedError

    def _nhp(self, d):
        return self._collect(d, 0, [], [], [])

    # ---- protocol ---------------------------------------------------------------------------
    def get_params_shape(self, x: Tensor) -> List[int]:
        return _params_shape(x, self._nhp(x.shape[-1]))

    def _batches(self, hp, x, xp=None):
        nhp = self._nhp(x.shape[-1])
        assert hp.shape[-1] == nhp  # covar.py:52,66,131,171
        hb = hp.reshape(-1, nhp)
        xb = x.reshape(-1, x.shape[-2], x.shape[-1])
        xpb = None if xp is None else xp.reshape(-1, xp.shape[-2], xp.shape[-1])
        nb = max(hb.shape[0], xb.shape[0], 1 if xpb is None else xpb.shape[0])
        for t in (hb, xb, xpb):
            if t is not None and t.shape[0] not in (1, nb):
                raise RuntimeError("batch dimensions of hp / x / xp do not broadcast")
        return hb, xb, xpb, nb

    def kernel(self, hp: Tensor, x: Tensor, xp: Tensor = None) -> Tensor:
        ops = get_ops()
        hb, xb, xpb, nb = self._batches(hp, x, xp)
        n, d = xb.shape[-2:]
        spec, _ = spec_of(self, d)
        dt = x.dtype if x.dtype in (torch.float32, torch.float64) else torch.float64
        hd = ops.to_device(hb, torch.float64)
        xd = ops.to_device(xb, dt)
        xpd = None if xpb is None else ops.to_device(xpb, dt)
        outs = []
        for b in range(nb):
            hpb = hd[b % hd.shape[0]]
            xr = xd[b % xd.shape[0]]
            if xpd is None:
                buf = ops.empty(pad_to(n, 64), pad_to(n, 64), dtype=dt)
                ops.kernel_build(spec, hpb, xr, None, buf)
                outs.append(buf[:n, :n])
            else:
                xq = xpd[b % xpd.shape[0]]
                m = xq.shape[0]
                buf = ops.empty(pad_to(m, 64), pad_to(n, 64), dtype=dt)
                ops.kernel_build(spec, hpb, xq, xr, buf)     # rows = test points (covar.py:152-161)
                outs.append(buf[:m, :n])
        res = torch.stack(outs) if nb > 1 else outs[0].contiguous()
        return res.to(x.device)

    def kernel_and_grad(self, hp: Tensor, x: Tensor) -> List[Tensor]:
        ops = get_ops()
        hb, xb, _, nb = self._batches(hp, x)
        n, d = xb.shape[-2:]
        spec, nhp = spec_of(self, d)
        dt = x.dtype if x.dtype in (torch.float32, torch.float64) else torch.float64
        hd = ops.to_device(hb, torch.float64)
        xd = ops.to_device(xb, dt)
        ks, dks = [], []
        for b in range(nb):
            hpb, xr = hd[b % hd.shape[0]], xd[b % xd.shape[0]]
            buf = ops.empty(pad_to(n, 64), pad_to(n, 64), dtype=dt)
            ops.kernel_build(spec, hpb, xr, None, buf)
            ks.append(buf[:n, :n])
            dks.append(ops.kernel_grad_build(spec, hpb, xr, ops.empty(nhp, n, n, dtype=dt)))
        k = torch.stack(ks) if nb > 1 else ks[0].contiguous()
        dk = torch.stack(dks) if nb > 1 else dks[0]
        return [k.to(x.device), dk.to(x.device)]


class Squared_exponential(_DeviceKernel):
    """ARD squared exponential, K(x,x') = sig^2 exp(-|(x-x').ls|^2) (PyGPR/covar.py:84-206)."""

    _kind = _lib.PG_KIND_RBF

    def _collect(self, d, base, kinds, offs, noise):
        kinds.append(self._kind)
        offs.append(base)
        return d + 1

    def init_params(self, x: Tensor) -> Tensor:  # covar.py:96-100
        return torch.ones(self.get_params_shape(x), dtype=torch.float64)

    def distance(self, x: Tensor, xp: Tensor = None) -> Tensor:
        """Squared Euclidean distances (covar.py:102-127): x [(b), n, d] -> [(b), n, n]; with xp [(b), m, d] the rows
        are the test points, [(b), m, n]; a batch of one is squeezed away like the reference does.  Evaluated on the
        device by the covariance tile kernel as direct sums of squared differences (the reference expands
        -2 x x'^T + |x|^2 + |x'|^2: same value to rounding, but this one is exactly symmetric with a zero diagonal)."""
        ops = get_ops()
        xb = x.reshape(-1, x.shape[-2], x.shape[-1])
        xpb = None if xp is None else xp.reshape(-1, xp.shape[-2], xp.shape[-1])
        nb = max(xb.shape[0], 1 if xpb is None else xpb.shape[0])
        for t in (xb, xpb):
            if t is not None and t.shape[0] not in (1, nb):
                raise RuntimeError("batch dimensions of x / xp do not broadcast")
        dt = x.dtype if x.dtype in (torch.float32, torch.float64) else torch.float64
        xd = ops.to_device(xb, dt)
        xpd = None if xpb is None else ops.to_device(xpb, dt)
        n = xb.shape[1]
        outs = []
        for b in range(nb):
            xr = xd[b % xd.shape[0]]
            if xpd is None:
                buf = ops.empty(pad_to(n, 64), pad_to(n, 64), dtype=dt)
                ops.sqdist(xr, None, buf)
                outs.append(buf[:n, :n])
            else:
                xq = xpd[b % xpd.shape[0]]
                m = xq.shape[0]
                buf = ops.empty(pad_to(m, 64), pad_to(n, 64), dtype=dt)
                ops.sqdist(xq, xr, buf)
                outs.append(buf[:m, :n])
        res = torch.stack(outs) if nb > 1 else outs[0].contiguous()
        return res.to(x.device)


class Matern52(Squared_exponential):
    """Matern-5/2 with the SE parameterisation: r = |(x-x').ls|,
    K = sig^2 (1 + sqrt5 r + 5 r^2/3) exp(-sqrt5 r).  Not in the reference."""

    _kind = _lib.PG_KIND_MATERN52


class White_noise(_DeviceKernel):
    """Gaussian noise sigma_n^2 I (PyGPR/covar.py:209-269)."""

    def _collect(self, d, base, kinds, offs, noise):
        noise.append(base)
        return 1

    def init_params(self, x: Tensor) -> Tensor:  # covar.py:221-225
        return 1e-4 * torch.ones(self.get_params_shape(x), dtype=torch.float64)

    def kernel(self, hp: Tensor, x: Tensor, xp: Tensor = None) -> Tensor:
        if xp is not None:
            return torch.tensor(0)  # covar.py:243
        return super().kernel(hp, x)


class Compose(_DeviceKernel):
    """Sum of covariance kernels (PyGPR/covar.py:28-81)."""

    def __init__(self, covars: Sequence[Covar]) -> None:
        self.covars = covars

    def _collect(self, d, base, kinds, offs, noise):
        used = 0
        for c in self.covars:
            used += c._collect(d, base + used, kinds, offs, noise)
        return used

    def init_params(self, x: Tensor) -> Tensor:  # covar.py:45-48
        return torch.cat([c.init_params(x) for c in self.covars], dim=-1)

    def kernel(self, hp: Tensor, x: Tensor, xp: Tensor = None) -> Tensor:
        if xp is not None and not layout(self, x.shape[-1])[0]:
            assert hp.shape[-1] == self._nhp(x.shape[-1])
            return torch.tensor(0)  # a sum of White_noise children only
        return super().kernel(hp, x, xp)
