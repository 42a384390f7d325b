"""Device-backed counterpart of ``particles.distributions`` (hot-path subset):
``ProbDist`` (:215-251), ``LocScaleDist`` (:262-264), ``Normal`` (:267-285) and
``MvNormal`` (:888-1009) with ``rvs`` / ``logpdf`` evaluated by HIP kernels.

Parameters and arguments may be python scalars, numpy arrays or DeviceArrays;
results are numpy arrays unless a DeviceArray was passed in.  Draws come from
the device's counted Philox stream (``particles_amd.seed``) -- or, with
``rvs(size, z=...)``, from standard normals supplied by the caller, which is
how the parity tests replay the reference's own draws.
"""
import numpy as np
from numpy import random
import numpy.linalg as nla

from . import _lib
from ._lib import DeviceArray, check, lib

HALFLOG2PI = 0.5 * np.log(2.0 * np.pi)


class ProbDist:
    """Base class for probability distributions (distributions.py:215-251)."""

    dim = 1
    dtype = float

    def shape(self, size):
        if size is None:
            return None
        return (size,) if self.dim == 1 else (size, self.dim)

    def logpdf(self, x):
        raise NotImplementedError

    def pdf(self, x):
        return np.exp(self.logpdf(x))

    def rvs(self, size=None):
        raise NotImplementedError

    def ppf(self, u):
        raise NotImplementedError


class LocScaleDist(ProbDist):
    """Base class for location-scale distributions (distributions.py:259-264)."""

    def __init__(self, loc=0.0, scale=1.0):
        self.loc = loc
        self.scale = scale


def _strided(v, N):
    """(DeviceArray, element stride, is_device_input) for a scalar / (1,) / (N,) value."""
    if isinstance(v, DeviceArray):
        return v, (0 if v.size == 1 else 1), True
    if isinstance(v, (float, int, np.floating)):           # (the common case: a scalar parameter)
        return DeviceArray.scalar(v), 0, False
    a = np.asarray(v, dtype=np.float64).reshape(-1)
    if a.size == 1 and N != 1:
        return DeviceArray.scalar(a[0]), 0, False
    if a.size not in (1, N):
        raise ValueError("operands could not be broadcast together with shape (%d,)" % N)
    return DeviceArray.from_numpy(a), (0 if a.size == 1 else 1), False


def _bsize(*vals):
    n = 1
    for v in vals:
        s = v.size if isinstance(v, DeviceArray) else np.asarray(v).size
        n = max(n, s)
    return n


class Normal(LocScaleDist):
    """N(loc, scale^2) distribution (distributions.py:267-285)."""

    def rvs(self, size=None, z=None):
        """``random.normal(loc, scale, size)`` = loc + scale * z  (:270-271)."""
        N = _bsize(self.loc, self.scale) if size is None else int(size)
        loc, ls, d1 = _strided(self.loc, N)
        sc, ss, d2 = _strided(self.scale, N)
        if z is None and _lib.RNG_MODE[0] == "numpy":
            z = random.standard_normal(N)      # random.normal(loc, scale, N) draws exactly these
        zd = None
        if z is not None:
            zd = z if isinstance(z, DeviceArray) else DeviceArray.from_numpy(
                np.asarray(z, dtype=np.float64).reshape(-1))
        out = DeviceArray((N,))
        check(lib().smc_normal_rvs(out.ctx.h, loc.ptr, ls, sc.ptr, ss,
                                   zd.ptr if zd is not None else None,
                                   _lib.next_counter(), N, out.ptr))
        if d1 or d2 or isinstance(z, DeviceArray) or _lib.RESIDENT[0]:
            return out
        r = out.get()
        return r if size is not None or N > 1 else r[0]

    def logpdf(self, x):
        """scipy.stats.norm.logpdf(x, loc, scale)  (:273-274)."""
        N = _bsize(self.loc, self.scale, x)
        xd, xs, d0 = _strided(x, N)
        loc, ls, d1 = _strided(self.loc, N)
        sc, ss, d2 = _strided(self.scale, N)
        out = DeviceArray((N,))
        check(lib().smc_normal_logpdf(out.ctx.h, xd.ptr, xs, loc.ptr, ls, sc.ptr, ss, N, out.ptr))
        return out if (d0 or d1 or d2 or _lib.RESIDENT[0]) else out.get()

    def ppf(self, u):
        """scipy.stats.norm.ppf(u, loc, scale) = ndtri(u) * scale + loc  (:276-277)."""
        N = _bsize(self.loc, self.scale, u)
        ud, us, d0 = _strided(u, N)
        loc, ls, d1 = _strided(self.loc, N)
        sc, ss, d2 = _strided(self.scale, N)
        out = DeviceArray((N,))
        check(lib().smc_normal_ppf(out.ctx.h, ud.ptr, us, loc.ptr, ls, sc.ptr, ss, N, out.ptr))
        return out if (d0 or d1 or d2 or _lib.RESIDENT[0]) else out.get()


class Poisson(ProbDist):
    """Poisson(rate) distribution (distributions.py:519-532).  ``logpdf`` is evaluated on the
    device for per-particle rates; ``rvs`` (only used to simulate data) draws on the host with
    numpy's generator, as the reference does."""
    dtype = "int64"

    def __init__(self, rate=1.0):
        self.rate = rate

    def rvs(self, size=None):
        rate = self.rate.get() if isinstance(self.rate, DeviceArray) else self.rate
        return random.poisson(rate, size=size)                                   # :525-526

    def logpdf(self, x):
        """scipy.stats.poisson.logpmf(x, rate)  (:528-529)."""
        N = _bsize(self.rate, x)
        xd, xs, d0 = _strided(np.asarray(x, dtype=np.float64) if not isinstance(x, DeviceArray) else x, N)
        rt, rs, d1 = _strided(self.rate, N)
        out = DeviceArray((N,))
        check(lib().smc_poisson_logpmf(out.ctx.h, xd.ptr, xs, rt.ptr, rs, N, out.ptr))
        return out if (d0 or d1 or _lib.RESIDENT[0]) else out.get()


class Dirac(ProbDist):
    """Dirac mass (distributions.py:454-472): the deterministic components of a product law (BearingsOnly's
    velocities).  Host arrays in, host arrays out -- there is nothing to compute."""

    def __init__(self, loc=0.0):
        self.loc = loc

    def rvs(self, size=None):
        if isinstance(self.loc, np.ndarray):
            return self.loc.copy()
        if isinstance(self.loc, DeviceArray):
            return self.loc.get()
        return np.full(1 if size is None else size, self.loc)

    def logpdf(self, x):
        loc = self.loc.get() if isinstance(self.loc, DeviceArray) else self.loc
        xx = x.get() if isinstance(x, DeviceArray) else x
        return np.where(xx == loc, 0.0, -np.inf)

    def ppf(self, u):
        return self.rvs(size=u.shape[0])


class IndepProd(ProbDist):
    """Product of independent univariate distributions (distributions.py:1066-1106): inputs and
    outputs of shape (N, d) -- numpy arrays, or device arrays (``set_resident``), whose columns
    are split / joined on the device."""

    def __init__(self, *dists):
        self.dists = dists
        self.dim = len(dists)
        self.dtype = "int64" if all(d.dtype == "int64" for d in dists) else float

    def logpdf(self, x):
        return sum([d.logpdf(x[..., i]) for i, d in enumerate(self.dists)])       # :1101-1102

    def rvs(self, size=None):
        return np.stack([d.rvs(size=size) for d in self.dists], axis=1)          # :1105-1106

    def ppf(self, u):
        return np.stack([d.ppf(u[..., i]) for i, d in enumerate(self.dists)], axis=1)   # :1108-1109


class MvNormal(ProbDist):
    """Multivariate Normal distribution (distributions.py:888-1009).

    ``loc``: (d,) or (N,d); ``scale``: scalar; ``cov``: (d,d).  (Per-component /
    per-particle ``scale`` arrays of the reference are outside the hot path.)
    """

    def __init__(self, loc=0.0, scale=1.0, cov=None):
        self.loc = loc
        self.scale = scale
        self.cov = np.eye(np.shape(loc)[-1]) if cov is None else cov
        err_msg = "MvNormal: argument cov must be a (d, d) pos. definite matrix"
        try:
            self.L = nla.cholesky(self.cov)  # lower triangle (:937)
        except nla.LinAlgError:
            raise ValueError(err_msg)
        assert self.cov.shape == (self.dim, self.dim), err_msg
        # a per-particle / per-component scale (distributions.py:925-929; MVStochVol's observation law): the kernels take a
        # scalar, so the array form standardises on the host side of the operator -- (x - loc) / scale through the unit-scale
        # law, minus sum(log scale) -- the reference's own lines :949-959 regrouped
        self._array_scale = np.ndim(scale) != 0

    @property
    def dim(self):
        return self.cov.shape[-1]

    def _loc(self, N):
        if isinstance(self.loc, DeviceArray):
            return self.loc, (1 if self.loc.size == self.dim else N), True
        a = np.asarray(self.loc, dtype=np.float64)
        if a.ndim == 0:
            a = np.full(self.dim, float(a))
        a = np.ascontiguousarray(a.reshape(-1, self.dim))
        if a.shape[0] not in (1, N):
            raise ValueError("MvNormal: loc has %d rows, expected 1 or %d" % (a.shape[0], N))
        return DeviceArray.from_numpy(a), a.shape[0], False

    def _unit(self):
        return MvNormal(loc=np.zeros(self.dim), scale=1.0, cov=self.cov)

    def rvs(self, size=None, z=None):
        """loc + scale * (Z @ L.T), Z ~ N(0, I)  (:946-947, :961-969)."""
        if self._array_scale:
            sc = np.asarray(self.scale.get() if isinstance(self.scale, DeviceArray) else self.scale, dtype=np.float64)
            lo = np.asarray(self.loc.get() if isinstance(self.loc, DeviceArray) else self.loc, dtype=np.float64)
            N = int(size) if size is not None else np.broadcast(lo, sc).shape[0]
            zl = self._unit().rvs(size=N, z=z)                  # Z @ L.T
            zl = zl.get() if isinstance(zl, DeviceArray) else zl
            return lo + sc * zl
        if size is None:
            sh = np.shape(self.loc) if not isinstance(self.loc, DeviceArray) else self.loc.shape
            N = sh[0] if len(sh) == 2 else 1
        else:
            N = int(size)
        loc, rows, dev = self._loc(N)
        if z is None and _lib.RNG_MODE[0] == "numpy":
            z = random.standard_normal((N, self.dim))      # stats.norm.rvs(size=(N, d)) (:968)
        zd = None
        if z is not None:
            zd = z if isinstance(z, DeviceArray) else DeviceArray.from_numpy(
                np.asarray(z, dtype=np.float64).reshape(N, self.dim))
        out = DeviceArray((N, self.dim))
        La, Lp = _lib.host_dbl(self.L)  # keep La alive across the call
        check(lib().smc_mvn_rvs(out.ctx.h, loc.ptr, rows, float(self.scale), Lp,
                                zd.ptr if zd is not None else None,
                                _lib.next_counter() if zd is None else 0,      # (a sub-stream only when it draws)
                                N, self.dim, out.ptr))
        return out if (dev or isinstance(z, DeviceArray) or _lib.RESIDENT[0]) else out.get()

    def ppf(self, u):
        """linear_transform(norm.ppf(u)) for u (N, du) (:970-981); du = dim only (the partly
        degenerate case du < dim of the reference is not built)."""
        ud, host = _lib.as_device(u)
        N, du = ud.shape
        if du != self.dim:
            raise NotImplementedError("MvNormal.ppf: u must have dim = %d columns" % self.dim)
        z = DeviceArray((N, du), np.float64, ud.ctx)
        zero = DeviceArray.from_numpy(np.array([0.0, 1.0]), context=ud.ctx)
        check(lib().smc_normal_ppf(ud.ctx.h, ud.ptr, 1, zero.ptr, 0, _lib.c_vp(zero.ptr.value + 8), 0,
                                   N * du, z.ptr))
        x = self.rvs(size=N, z=z)
        return x if (not host or _lib.RESIDENT[0] or isinstance(self.loc, DeviceArray)) else x.get()

    def logpdf(self, x):
        """:949-959."""
        if self._array_scale:
            sc = np.asarray(self.scale.get() if isinstance(self.scale, DeviceArray) else self.scale, dtype=np.float64)
            lo = np.asarray(self.loc.get() if isinstance(self.loc, DeviceArray) else self.loc, dtype=np.float64)
            xx = np.asarray(x.get() if isinstance(x, DeviceArray) else x, dtype=np.float64)
            base = self._unit().logpdf(np.ascontiguousarray((xx - lo) / sc))
            base = base.get() if isinstance(base, DeviceArray) else base
            return base - np.sum(np.log(sc), axis=-1)
        if isinstance(x, DeviceArray):
            xd, xrows, dx = x, x.size // self.dim, True
        else:
            xa = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, self.dim))
            xd, xrows, dx = DeviceArray.from_numpy(xa), xa.shape[0], False
        lsh = self.loc.shape if isinstance(self.loc, DeviceArray) else np.shape(self.loc)
        N = max(xrows, lsh[0] if len(lsh) == 2 else 1)
        loc, rows, dl = self._loc(N)
        out = DeviceArray((N,))
        La, Lp = _lib.host_dbl(self.L)  # keep La alive across the call
        check(lib().smc_mvn_logpdf(out.ctx.h, xd.ptr, xrows, loc.ptr, rows, float(self.scale),
                                   Lp, N, self.dim, out.ptr))
        return out if (dx or dl or _lib.RESIDENT[0]) else out.get()
