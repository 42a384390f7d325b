"""Device-backed counterpart of ``particles.resampling`` (hot-path subset).

Same names, arguments and error behaviour as particles/resampling.py for:
``Weights`` (:191-244), ``exp_and_normalise`` (:138), ``essl`` (:166),
``log_sum_exp`` (:247), ``log_mean_exp`` (:291), ``wmean_and_var`` (:320),
``rs_funcs`` / ``resampling_scheme`` / ``resampling`` (:445-481),
``inverse_cdf`` (:484-509), ``uniform_spacings`` (:512-537), ``multinomial``
(:540), ``stratified`` (:599), ``systematic`` (:606).

Arrays may be numpy arrays (copied to HBM, result copied back -- drop-in
behaviour) or ``DeviceArray`` (stay resident).  All arithmetic runs in the HIP
kernels of libsmc_hip.so; nothing here computes on the CPU.

Uniform draws: with numpy inputs the uniforms are taken from the numpy global
generator exactly as the reference does (same count, same order), so a run
seeded with ``numpy.random.seed`` selects the same ancestors as the reference
(up to certified near-ties of the CDF, DESIGN.md "Q62 contract").  With
``set_rng('philox')`` or DeviceArray inputs they come from the device's counted
Philox stream instead.
"""
import ctypes
import functools

import numpy as np
from numpy import random

from . import _lib
from ._lib import DeviceArray, as_device, check, lib



def set_rng(mode):
    """'numpy' (reference-compatible draws on the host) or 'philox' (device)."""
    if mode not in ("numpy", "philox"):
        raise ValueError("rng mode must be 'numpy' or 'philox'")
    _lib.RNG_MODE[0] = mode


def _normalise(lw_dev, want_W=True):
    N = lw_dev.size
    W = DeviceArray((N,)) if want_W else None
    out = (ctypes.c_double * 4)()
    check(lib().smc_lse_normalise(lw_dev.ctx.h, lw_dev.ptr, N, W.ptr if W else None, out))
    return W, out[0], out[1], (out[2], out[3])


def exp_and_normalise(lw):
    """W = exp(lw) / sum(exp(lw))  (resampling.py:138-163)."""
    d, host = as_device(np.array(lw, dtype=float) if not isinstance(lw, DeviceArray) else lw)
    W, _, _, _ = _normalise(d)
    return W.get() if (host and not _lib.RESIDENT[0]) else W


def essl(lw):
    """ESS from log-weights (resampling.py:166-188)."""
    d, _ = as_device(np.array(lw, dtype=float) if not isinstance(lw, DeviceArray) else lw)
    _, _, ess, _ = _normalise(d, want_W=False)
    return ess


def log_sum_exp(v):
    """log(sum(exp(v)))  (resampling.py:247-270)."""
    d, _ = as_device(np.array(v, dtype=float) if not isinstance(v, DeviceArray) else v)
    _, _, _, (m, s) = _normalise(d, want_W=False)
    return m + np.log(s)


def log_sum_exp_ab(a, b):
    """log(e^a + e^b) for two scalars (resampling.py:273-288); host scalar code."""
    if a > b:
        return a + np.log1p(np.exp(b - a))
    return b + np.log1p(np.exp(a - b))


def log_mean_exp(v, W=None):
    """log of the (weighted) mean of exp(v)  (resampling.py:291-317)."""
    d, _ = as_device(np.array(v, dtype=float) if not isinstance(v, DeviceArray) else v)
    if W is None:
        _, log_mean, _, _ = _normalise(d, want_W=False)
        return log_mean
    Wd, _ = as_device(W)
    out = ctypes.c_double()
    check(lib().smc_log_wmean_exp(d.ctx.h, d.ptr, Wd.ptr, d.size, ctypes.byref(out)))
    return out.value


def wmean_and_var(W, x):
    """Component-wise weighted mean and variance (resampling.py:320-338)."""
    Wd, _ = as_device(W)
    xd, _ = as_device(x)
    N = Wd.size
    dd = xd.size // N
    out = (ctypes.c_double * (2 * dd))()
    check(lib().smc_wmean_var(Wd.ctx.h, Wd.ptr, xd.ptr, N, dd, out))
    o = np.array(out[:])
    if xd.ndim == 1:
        return {"mean": o[0], "var": o[1]}
    return {"mean": o[:dd], "var": o[dd:]}


def wmean_and_cov(W, x):
    """Weighted mean and covariance matrix (resampling.py:341-358): ``(mean, cov)``, cov = np.cov(x.T, aweights=W,
    ddof=0); a 0-d array for one-dimensional data, as numpy returns it."""
    Wd, _ = as_device(W)
    xd, _ = as_device(x)
    N = Wd.size
    dd = xd.size // N
    if dd > 32:                                   # (beyond the kernel's tile: numpy on the host, the reference's own lines)
        xh = xd.get().reshape(N, dd)
        Wh = Wd.get()
        return np.average(xh, weights=Wh, axis=0), np.cov(xh.T, aweights=Wh, ddof=0)
    out = (ctypes.c_double * (dd + dd * dd))()
    check(lib().smc_wmean_cov(Wd.ctx.h, Wd.ptr, xd.ptr, N, dd, out))
    o = np.array(out[:])
    if xd.ndim == 1:
        return o[0], np.array(o[1])
    return o[:dd], o[dd:].reshape(dd, dd)


def wmean_and_var_str_array(W, x):
    """Weighted mean and variance of each component of a structured array (resampling.py:361-380)."""
    m = np.empty(shape=x.shape[1:], dtype=x.dtype)
    v = np.empty_like(m)
    for p in x.dtype.names:
        m[p], v[p] = wmean_and_var(W, np.ascontiguousarray(x[p])).values()
    return {"mean": m, "var": v}


def wquantiles_str_array(W, x, alphas=(0.25, 0.50, 0.75)):
    """Quantiles of weighted data stored in a structured array, one entry per field (resampling.py:420-442; the
    reference's default ``alphas=(0.25, 0.50, 0, 75)`` is a typo for these three)."""
    return {p: wquantiles(W, np.ascontiguousarray(x[p]), alphas) for p in x.dtype.names}


def wquantiles(W, x, alphas=(0.25, 0.50, 0.75)):
    """Quantiles for weighted data (resampling.py:381-417): ``(k,)`` for ``x`` of shape
    ``(N,)``, ``(d, k)`` for ``(N, d)``."""
    Wd, _ = as_device(W)
    xd, _ = as_device(x)
    N = Wd.size
    dd = xd.size // N
    al = np.ascontiguousarray(alphas, dtype=np.float64)
    out = np.empty((dd, al.size))
    check(lib().smc_wquantiles(Wd.ctx.h, Wd.ptr, xd.ptr, N, dd,
                               al.ctypes.data_as(_lib.P(_lib.c_dbl)), al.size,
                               out.ctypes.data_as(_lib.P(_lib.c_dbl))))
    if xd.ndim == 1:
        return list(out[0])
    return out


class Weights:
    """N log-weights with their normalised weights and ESS (resampling.py:191-244).

    ``lw`` is a numpy array (then ``W`` is numpy too) or a DeviceArray.  As in the
    reference, NaN log-weights are replaced by -inf in the caller's array
    (:220), objects are to be treated as immutable and ``add`` returns a new one.
    """

    def __init__(self, lw=None):
        self.lw = lw
        if lw is not None:
            if isinstance(lw, DeviceArray):
                d = lw
            else:
                self.lw[np.isnan(self.lw)] = -np.inf        # :220, caller's array
                d = DeviceArray.from_numpy(self.lw)
            W, self.log_mean, self.ESS, _ = _normalise(d)
            self.W = W if isinstance(lw, DeviceArray) else W.get()

    @property
    def N(self):
        return 0 if self.lw is None else self.lw.shape[0]

    def add(self, delta):
        """lw <- lw + delta (resampling.py:232-244)."""
        if self.lw is None:
            return self.__class__(lw=delta)
        if isinstance(self.lw, DeviceArray) or isinstance(delta, DeviceArray):
            a = self.lw if isinstance(self.lw, DeviceArray) else DeviceArray.from_numpy(self.lw)
            return self.__class__(lw=a + delta)             # element-wise add on the device
        return self.__class__(lw=self.lw + delta)


####################
# Resampling schemes
####################

rs_funcs = {}  # populated by the decorator below

rs_doc = """\

    Parameters
    ----------
    W : (N,) ndarray or DeviceArray
        normalized weights (>=0, sum to one)
    M : int, optional (set to N if missing)
        number of resampled points.

    Returns
    -------
    (M,) int64 ndarray (DeviceArray if W was one)
     M ancestor variables, drawn from range 0, ..., N-1
"""


def resampling_scheme(func):
    """Decorator for resampling schemes (resampling.py:464-474)."""

    @functools.wraps(func)
    def modif_func(W, M=None):
        M = W.shape[0] if M is None else M
        return func(W, M)

    rs_funcs[func.__name__] = modif_func
    modif_func.__doc__ = (func.__doc__ or "") + rs_doc
    return modif_func


def resampling(scheme, W, M=None):
    """resampling.py:477-481."""
    try:
        return rs_funcs[scheme](W, M=M)
    except KeyError:
        raise ValueError(f"{scheme} is not a valid resampling scheme")


STRICT = [False]


def set_strict(flag=True):
    """Resample with the reference's own CDF -- S_j accumulated left to right in fp64
    (resampling.py:500-509) -- instead of the exact integer CDF: ``inverse_cdf`` and the schemes built on
    it then return the reference's ancestors bit for bit for identical ``(su, W)``, at milliseconds per
    call at N = 2^20 (one wavefront walks the weights: the order of the additions is the result)."""
    STRICT[0] = bool(flag)


class strict_mode:
    """``with strict_mode(flag):`` -- ``set_strict(flag)`` for the block (what ``SMC(strict_ancestors=True)`` wraps the
    resampling calls of its template-method step in, so that the option means the same on every path)."""

    def __init__(self, flag=True):
        self.flag = bool(flag)

    def __enter__(self):
        self.saved, STRICT[0] = STRICT[0], self.flag
        return self

    def __exit__(self, *exc):
        STRICT[0] = self.saved
        return False


def inverse_cdf(su, W, strict=None):
    """Inverse CDF algorithm for a finite distribution (resampling.py:484-509).

    su: M sorted points in [0,1]; returns, for each, the smallest index j with
    su[n] <= CDF_j (clamped to N-1 where the reference would run off the end).
    strict (default: ``set_strict``'s value): the sequential fp64 CDF of the reference, literally.
    """
    sud, host = as_device(su)
    Wd, hostW = as_device(W)
    A = DeviceArray((sud.size,), np.int64)
    fn = lib().smc_inverse_cdf_strict if (STRICT[0] if strict is None else strict) else lib().smc_inverse_cdf
    check(fn(Wd.ctx.h, sud.ptr, Wd.ptr, sud.size, Wd.size, A.ptr))
    return A.get() if (host and hostW and not _lib.RESIDENT[0]) else A


def uniform_spacings(N):
    """N ordered uniforms in O(N) (resampling.py:512-537).

    'numpy' mode: the reference's expression on the host generator's draws
    (that part is RNG plumbing, not the hot path); 'philox' mode: drawn and
    scanned on the device, returned as a DeviceArray.
    """
    if _lib.RNG_MODE[0] == "numpy":
        z = np.cumsum(-np.log(random.rand(N + 1)))
        return z[:-1] / z[-1]
    su = DeviceArray((N,))
    check(lib().smc_uniform_spacings(su.ctx.h, N, _lib.next_counter(), su.ptr))
    return su


def _resample(scheme, W, M):
    if STRICT[0]:
        # the reference's expressions for the sorted uniforms (:536-537, :602, :609) on its draws (or the
        # device's), then the reference's CDF
        if _lib.RNG_MODE[0] == "numpy":
            draw = random.rand
        else:
            def draw(k):
                u = DeviceArray((max(k, 1),))
                check(lib().smc_uniform(u.ctx.h, _lib.next_counter(), max(k, 1), u.ptr))
                return u.get()[:k]
        if scheme == "systematic":
            su = (draw(1) + np.arange(M)) / M
        elif scheme == "stratified":
            su = (draw(M) + np.arange(M)) / M
        else:
            su = uniform_spacings(M)
        return inverse_cdf(su, W, strict=True)
    Wd, host = as_device(W)
    A = DeviceArray((M,), np.int64)
    u = None
    if _lib.RNG_MODE[0] == "numpy":
        # the reference's draws, in the reference's order (:536, :602, :609)
        if scheme == "systematic":
            u = DeviceArray.from_numpy(random.rand(1))
        elif scheme == "stratified":
            u = DeviceArray.from_numpy(random.rand(M))
        else:
            u = DeviceArray.from_numpy(uniform_spacings(M))
    check(lib().smc_resample(Wd.ctx.h, _lib.SCHEMES[scheme], Wd.ptr, Wd.size, M,
                             u.ptr if u is not None else None, _lib.next_counter(), A.ptr))
    return A.get() if (host and not _lib.RESIDENT[0]) else A


@resampling_scheme
def multinomial(W, M):
    """Multinomial resampling (resampling.py:540-558); output is ordered."""
    return _resample("multinomial", W, M)


@resampling_scheme
def stratified(W, M):
    """Stratified resampling (resampling.py:599-603)."""
    return _resample("stratified", W, M)


@resampling_scheme
def systematic(W, M):
    """Systematic resampling (resampling.py:606-610)."""
    return _resample("systematic", W, M)


@resampling_scheme
def residual(W, M):
    """Residual resampling (resampling.py:611-626): ``floor(M W)`` copies of every
    particle, then a multinomial draw of the remaining ``M - sum floor(M W)`` on the
    residuals -- both parts on the device."""
    Wd, host = as_device(W)
    N = Wd.size
    r = DeviceArray((N,))
    sip = _lib.c_i64()
    check(lib().smc_residual_split(Wd.ctx.h, Wd.ptr, N, M, r.ptr, ctypes.byref(sip)))
    sres = M - sip.value
    A = DeviceArray((M,), np.int64)
    su = None
    if sres > 0:                      # multinomial(res / sres, M=sres): its draws (:536)
        su = uniform_spacings(sres)
        su = su if isinstance(su, DeviceArray) else DeviceArray.from_numpy(su)
    check(lib().smc_residual_ancestors(Wd.ctx.h, Wd.ptr, r.ptr, N, M, sip.value,
                                       su.ptr if su is not None else None, A.ptr))
    return A.get() if (host and not _lib.RESIDENT[0]) else A


@resampling_scheme
def ssp(W, M):
    """SSP resampling (resampling.py:628-678): offspring numbers floor(M W) or floor(M W)+1,
    consistent (Gerber, Chopin & Whiteley 2019).  Inherently sequential: one lane of the
    device walks the chain, the weights stay resident."""
    Wd, host = as_device(W)
    N = Wd.size
    if _lib.RNG_MODE[0] == "numpy":
        u = DeviceArray.from_numpy(random.rand(N - 1)) if N > 1 else None        # :649
    else:
        u = DeviceArray((max(N - 1, 1),))
        check(lib().smc_uniform(u.ctx.h, _lib.next_counter(), max(N - 1, 1), u.ptr))
    A = DeviceArray((M,), np.int64)
    check(lib().smc_resample_ssp(Wd.ctx.h, Wd.ptr, u.ptr if u is not None else None, N, M, A.ptr))
    return A.get() if (host and not _lib.RESIDENT[0]) else A


@resampling_scheme
def killing(W, M):
    """Killing resampling (resampling.py:680-697): keep particle i with probability
    ``W[i] / W.max()``, otherwise replace it by a multinomial draw.  Requires M = N."""
    Wd, host = as_device(W)
    N = Wd.size
    if M != N:
        raise ValueError("killing resampling defined only for M=N")
    if _lib.RNG_MODE[0] == "numpy":
        u = DeviceArray.from_numpy(random.rand(N))                 # :692
    else:
        u = DeviceArray((N,))
        check(lib().smc_uniform(u.ctx.h, _lib.next_counter(), N, u.ptr))
    killed = DeviceArray(((N + 7) // 8,))           # N bytes of flags
    nk = _lib.c_i64()
    check(lib().smc_killing_split(Wd.ctx.h, Wd.ptr, u.ptr, N, killed.ptr, ctypes.byref(nk)))
    Am = None
    if _lib.RNG_MODE[0] == "numpy":              # multinomial(W, nkilled) on the reference's draws (:695);
        su = uniform_spacings(nk.value)   # with nkilled = 0 it still consumes rand(1)
        if nk.value > 0:
            Am = inverse_cdf(DeviceArray.from_numpy(su), Wd)
    elif nk.value > 0:
        Am = multinomial(Wd, M=nk.value)
    A = DeviceArray((N,), np.int64)
    check(lib().smc_killing_ancestors(Wd.ctx.h, killed.ptr, Am.ptr if Am is not None else None,
                                      N, A.ptr))
    return A.get() if (host and not _lib.RESIDENT[0]) else A


def multinomial_once(W):
    """One draw from the discrete distribution W (resampling.py:574-596):
    ``np.searchsorted(np.cumsum(W), rand())``."""
    Wh = W.get() if isinstance(W, DeviceArray) else np.asarray(W)
    return int(np.searchsorted(np.cumsum(Wh), random.rand()))


def multinomial_iid(W, M=None):
    """Multinomial resampling, randomly permuted (resampling.py:561-571)."""
    A = multinomial(W, M=M)
    if isinstance(A, DeviceArray):
        A = A.get()
    random.shuffle(A)
    return A
