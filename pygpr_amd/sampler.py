"""Input samplers and expert partitioning with PyGPR's surface (reference: PyGPR/sampler.py).

Host-side input generation (not on the timed path): `UNIFORM`, `MATERN1` (a Matern type-I hard-core point process, NOT a
covariance kernel) draw from torch's global generator with the reference's call sequence, so the same seed gives the
same points.  The data-parallel part -- squared distances to the centres and the nearest-centre assignment that
builds the grBCM shards (`cluster_samples`, `partition`) -- runs on the GPU (`pg_sqdist_argmin`), with direct
differences instead of the reference's GEMM expansion (sampler.py:94-100).  `sample_gp` is dead code in the reference
(it calls `cov(x)`, sampler.py:122-137) and is not reproduced.
"""
import torch

from ._ops import get_ops


class UNIFORM(object):
    """Uniform points in the box [mins, maxs] (sampler.py:6-14)."""

    def __init__(self, seed):
        self.seed = seed

    def _box(self, n, mins, maxs):
        return mins + torch.rand(n, len(mins), dtype=torch.float64).mul_(maxs - mins)

    def sample(self, n, mins, maxs):
        torch.manual_seed(self.seed)
        return self._box(n, mins, maxs)


def _device_pair(x, c):
    ops = get_ops()
    dt = x.dtype if x.dtype == torch.float32 else torch.float64
    return ops, ops.to_device(x.reshape(-1, x.shape[-1]), dt), ops.to_device(c.reshape(-1, c.shape[-1]), dt)


def euclidean_dist(x, y):
    """Squared Euclidean distances [n, m] (sampler.py:94-100)."""
    ops, xd, yd = _device_pair(x, y)
    out = ops.empty(xd.shape[0], yd.shape[0], dtype=xd.dtype)
    ops.sqdist_argmin(xd, yd, dist=out)
    return out.to(x.device)


def nearest_centre(x, xc):
    """Index of the nearest centre for every point (the argmin of sampler.py:80-82,112-116), int64 on x's device."""
    ops, xd, cd = _device_pair(x, xc)
    idx = torch.empty(xd.shape[0], dtype=torch.int32, device=ops.device)
    ops.sqdist_argmin(xd, cd, idx=idx)
    return idx.to(device=x.device, dtype=torch.int64)


def cluster_samples(x, xc):
    """Split x [n, d] into nc equal shards by nearest centre (sampler.py:103-119); like the reference it only works
    when every centre attracts exactly n / nc points."""
    n, nc = x.shape[-2], xc.shape[-2]
    assert n % nc == 0
    ns = n // nc
    idx = nearest_centre(x, xc)
    xpart = torch.empty([nc, ns, x.shape[-1]], dtype=x.dtype, device=x.device)
    for i in range(nc):
        xpart[i, :, :] = x[idx == i, :]
    return xpart


class MATERN1(UNIFORM):
    """Hard-core (Matern type I) sampling of well-separated centres and Voronoi-style shards around them
    (sampler.py:17-91)."""

    def __init__(self, seed):
        super().__init__(seed)
        self.min_dist = None
        self.max_count = 5000

    def sample_repulsion(self, mins, maxs, min_dist):
        """Sequential rejection: keep a candidate when it is farther than min_dist (+1e-5) from every kept point; stop
        after max_count points or max_count consecutive rejections (sampler.py:23-48)."""
        torch.manual_seed(self.seed)
        dim = len(mins)
        kept = torch.empty([self.max_count, dim], dtype=torch.float64)
        kept[0, :] = self._box(1, mins, maxs)
        k, misses = 1, 1
        while k < self.max_count and misses < self.max_count:
            cand = self._box(1, mins, maxs)
            gap = (kept[:k, :] - cand).square_().sum(1).sqrt_()
            if bool(torch.all(gap.sub_(min_dist) > 1e-5)):
                kept[k, :] = cand
                k += 1
                misses = 0
            misses += 1
        return kept[:k, :]

    def sample(self, n, mins, maxs):
        vol = torch.prod(maxs - mins)
        min_dist = (vol / n) ** (1 / len(mins))
        xc = self.sample_repulsion(mins, maxs, min_dist)
        while xc.shape[0] < n:          # shrink the exclusion radius until n points fit (sampler.py:57-59)
            min_dist *= 0.9
            xc = self.sample_repulsion(mins, maxs, min_dist)
        self.min_dist = min_dist
        return xc[:n, :]

    def cluster_samples(self, xc, ns, mins, maxs):
        """ns points per centre out of 10 ns nc uniform candidates, by nearest centre (sampler.py:68-85)."""
        torch.manual_seed(self.seed)
        nc, dim = xc.shape
        x = self._box(10 * ns * nc, mins, maxs)
        idx = nearest_centre(x, xc)
        xpart = torch.empty([nc, ns, dim], dtype=torch.float64)
        for i in range(nc):
            xpart[i, :, :] = x[idx == i][:ns, :]
        return xpart

    def partition(self, nc, ns, mins, maxs):
        xc = self.sample(nc, mins, maxs)
        return self.cluster_samples(xc, ns, mins, maxs), xc


def sample_gp(*args, **kwargs):
    """Exported by the reference (PyGPR/__init__.py:6) but dead there: sampler.py:122-137 calls `cov(x)` on objects that
    are not callable ("dead" row).  Kept as a name for import compatibility."""
    raise NotImplementedError("sample_gp is legacy code that cannot run in the reference either (sampler.py:122-137)")
