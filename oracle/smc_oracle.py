"""NumPy restatement of the nchopin/particles SMC hot path (CPU oracle).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Nothing under
``particles_amd/`` imports this module.

Every function cites the reference lines it follows (paths relative to
/root/reference).  Third-party arithmetic the reference calls (numpy.random
legacy generator, numpy/scipy linear algebra) is *called*, not restated, since
the same numpy/scipy are present wherever the oracle runs.

Pinned by ``tests/golden/*.npz`` (outputs of the imported reference produced by
``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` checks the
oracle against them bit-for-bit.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import scipy.linalg as sla
import scipy.special as ssp_special

C_NORM = 0.9189385332046727  # scipy.stats._continuous_distns._norm_pdf_logC
HALFLOG2PI = 0.5 * np.log(2.0 * np.pi)  # particles/distributions.py:212

_HERE = os.path.dirname(os.path.abspath(__file__))
_CLIB = None


def clib():
    """The C half of the oracle (oracle/oracle.c), or None if not built."""
    global _CLIB
    if _CLIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            return None
        lib = ctypes.CDLL(path)
        i64, dp, ip, up = (ctypes.c_int64, ctypes.POINTER(ctypes.c_double),
                           ctypes.POINTER(ctypes.c_int64),
                           ctypes.POINTER(ctypes.c_uint64))
        lib.orc_inverse_cdf_seq.argtypes = [dp, dp, i64, i64, ip]
        lib.orc_inverse_cdf_seq.restype = i64
        lib.orc_inverse_cdf_q62.argtypes = [dp, dp, i64, i64, ip]
        lib.orc_inverse_cdf_q62.restype = None
        lib.orc_philox4x32_10.argtypes = [ctypes.POINTER(ctypes.c_uint32)] * 3
        lib.orc_philox4x32_10.restype = None
        lib.orc_toy_filter_philox.argtypes = [
            dp, i64, i64, ctypes.c_double, ctypes.c_double, ctypes.c_double,
            ctypes.c_double, ctypes.c_double, ctypes.c_uint64, dp]
        lib.orc_toy_filter_philox.restype = ctypes.c_double
        lib.orc_exp_nonpos_v.argtypes = [dp, i64, dp]
        lib.orc_exp_nonpos_v.restype = None
        lib.orc_tile_partials.argtypes = [dp, i64, dp, dp, dp, up]
        lib.orc_tile_partials.restype = None
        lib.orc_weights_pk.argtypes = [dp, i64, ctypes.c_double, dp]
        lib.orc_weights_pk.restype = None
        lib.orc_two_level_reduce.argtypes = [dp, dp, dp, i64, dp, dp, dp]
        lib.orc_two_level_reduce.restype = None
        lib.orc_inverse_cdf_2level.argtypes = [dp, i64, ctypes.c_int, dp, ip, dp]
        lib.orc_inverse_cdf_2level.restype = ctypes.c_int
        _CLIB = lib
    return _CLIB


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


# --------------------------------------------------------------------------
# Weights / log-sum-exp          (particles/resampling.py)
# --------------------------------------------------------------------------

class Weights:
    """resampling.py:191-244 (``Weights.__init__`` / ``add`` / ``N``)."""

    def __init__(self, lw=None):
        self.lw = lw
        if lw is not None:
            self.lw[np.isnan(self.lw)] = -np.inf          # :220 (in place!)
            m = self.lw.max()                               # :221
            w = np.exp(self.lw - m)                         # :222
            s = w.sum()                                     # :223
            self.log_mean = m + np.log(s / self.N)          # :224
            self.W = w / s                                  # :225
            self.ESS = 1.0 / np.sum(self.W ** 2)            # :226

    @property
    def N(self):
        return 0 if self.lw is None else self.lw.shape[0]   # :228-230

    def add(self, delta):
        if self.lw is None:                                 # :241-244
            return self.__class__(lw=delta)
        return self.__class__(lw=self.lw + delta)


def exp_and_normalise(lw):
    """resampling.py:138-163."""
    w = np.exp(lw - lw.max())
    return w / w.sum()


def essl(lw):
    """resampling.py:166-188."""
    w = np.exp(lw - lw.max())
    return (w.sum()) ** 2 / np.sum(w ** 2)


def log_sum_exp(v):
    """resampling.py:247-270."""
    m = v.max()
    return m + np.log(np.sum(np.exp(v - m)))


def log_mean_exp(v, W=None):
    """resampling.py:291-317."""
    m = v.max()
    V = np.exp(v - m)
    if W is None:
        return m + np.log(np.mean(V))
    return m + np.log(np.average(V, weights=W))


def wmean_and_var(W, x):
    """resampling.py:320-338."""
    m = np.average(x, weights=W, axis=0)
    m2 = np.average(x ** 2, weights=W, axis=0)
    return {"mean": m, "var": m2 - m ** 2}


def wmean_and_cov(W, x):
    """resampling.py:341-358."""
    m = np.average(x, weights=W, axis=0)
    cov = np.cov(x.T, aweights=W, ddof=0)
    return m, cov


def wmean_and_var_str_array(W, x):
    """resampling.py:361-380."""
    m = np.empty(shape=x.shape[1:], dtype=x.dtype)
    v = np.empty_like(m)
    for p in x.dtype.names:
        m[p], v[p] = wmean_and_var(W, x[p]).values()
    return {"mean": m, "var": v}


def wquantiles_str_array(W, x, alphas=(0.25, 0.50, 0.75)):
    """resampling.py:420-442."""
    return {p: wquantiles(W, x[p], alphas) for p in x.dtype.names}


# --------------------------------------------------------------------------
# Resampling                      (particles/resampling.py)
# --------------------------------------------------------------------------

def inverse_cdf(su, W):
    """resampling.py:484-509 -- sequential fp64 CDF, strict ``>`` compare.

    Uses the C restatement (oracle.c: orc_inverse_cdf_seq) when built, else the
    same loop in Python.  Raises IndexError where the pure-Python reference
    would (su[n] > sum(W) through round-off).
    """
    su = np.ascontiguousarray(su, dtype=np.float64)
    W = np.ascontiguousarray(W, dtype=np.float64)
    M, N = su.shape[0], W.shape[0]
    A = np.empty(M, dtype=np.int64)
    lib = clib()
    if lib is not None:
        rc = lib.orc_inverse_cdf_seq(
            _dp(su), _dp(W), M, N,
            A.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
        if rc != 0:
            raise IndexError("inverse_cdf: su exceeds the total weight")
        return A
    j = 0
    s = W[0]
    for n in range(M):
        while su[n] > s:
            j += 1
            s += W[j]
        A[n] = j
    return A


def uniform_spacings_from(u):
    """resampling.py:536-537 with the ``rand(N+1)`` draws passed in as ``u``."""
    z = np.cumsum(-np.log(u))
    return z[:-1] / z[-1]


def su_systematic(M, u):
    """resampling.py:609.  ``u`` = the ``rand(1)`` draw (shape (1,))."""
    return (u + np.arange(M)) / M


def su_stratified(M, u):
    """resampling.py:602.  ``u`` = the ``rand(M)`` draws."""
    return (u + np.arange(M)) / M


N_UNIFORMS = {"systematic": lambda M: 1, "stratified": lambda M: M,
              "multinomial": lambda M: M + 1}


def sorted_uniforms(scheme, M, u):
    if scheme == "systematic":
        return su_systematic(M, u)
    if scheme == "stratified":
        return su_stratified(M, u)
    if scheme == "multinomial":
        return uniform_spacings_from(u)
    raise ValueError("%s is not a valid resampling scheme" % scheme)  # :477-481


def resampling(scheme, W, M=None, u=None, rng=None, cdf="seq"):
    """resampling.py:477-481 + :540-558, :599-610.

    ``u`` are the uniforms the scheme consumes (1, M, or M+1 of them); if None
    they are drawn from ``rng`` (default: the numpy legacy global generator,
    like the reference).  ``cdf='q62'`` swaps the sequential fp64 CDF for the
    fixed-point contract of ``inverse_cdf_q62`` (what the HIP kernels use).
    """
    M = W.shape[0] if M is None else M
    if scheme not in N_UNIFORMS:
        raise ValueError("%s is not a valid resampling scheme" % scheme)
    if u is None:
        rng = LegacyRNG() if rng is None else rng
        u = rng.rand(N_UNIFORMS[scheme](M))
    su = sorted_uniforms(scheme, M, u)
    return inverse_cdf(su, W) if cdf == "seq" else inverse_cdf_q62(su, W)


def residual(W, M=None, rng=None, cdf="seq"):
    """resampling.py:611-626: floor(M W) copies, then multinomial on the residuals."""
    N = W.shape[0]
    M = N if M is None else M
    rng = LegacyRNG() if rng is None else rng
    A = np.empty(M, dtype=np.int64)
    MW = M * W
    intpart = np.floor(MW).astype(np.int64)
    sip = np.sum(intpart)
    res = MW - intpart
    sres = M - sip
    A[:sip] = np.arange(N).repeat(intpart)
    if sres > 0:
        A[sip:] = resampling("multinomial", res / sres, M=sres, rng=rng, cdf=cdf)
    return A


def ssp(W, M=None, rng=None):
    """resampling.py:628-678 (Srinivasan sampling process), statement by statement."""
    N = W.shape[0]
    M = N if M is None else M
    rng = LegacyRNG() if rng is None else rng
    MW = M * W
    nr_children = np.floor(MW).astype(np.int64)
    xi = MW - nr_children
    u = rng.rand(N - 1)
    i, j = 0, 1
    k = -1
    for k in range(N - 1):
        delta_i = min(xi[j], 1.0 - xi[i])
        delta_j = min(xi[i], 1.0 - xi[j])
        sum_delta = delta_i + delta_j
        pj = delta_i / sum_delta if sum_delta > 0.0 else 0.0
        if u[k] < pj:
            j, i = i, j
            delta_i = delta_j
        if xi[j] < 1.0 - xi[i]:
            xi[i] += delta_i
            j = k + 2
        else:
            xi[j] -= delta_i
            nr_children[i] += 1
            i = k + 2
    if np.sum(nr_children) == M - 1:
        last_ij = i if j == k + 2 else j
        if xi[last_ij] > 0.99:
            nr_children[last_ij] += 1
    if np.sum(nr_children) != M:
        raise ValueError("ssp resampling: wrong size for output")
    return np.arange(N).repeat(nr_children)


def killing(W, M=None, rng=None, cdf="seq"):
    """resampling.py:680-697."""
    N = W.shape[0]
    M = N if M is None else M
    if M != N:
        raise ValueError("killing resampling defined only for M=N")
    rng = LegacyRNG() if rng is None else rng
    killed = rng.rand(N) * W.max() >= W
    nkilled = killed.sum()
    A = np.arange(N)
    A[killed] = resampling("multinomial", W, M=nkilled, rng=rng, cdf=cdf)
    return A


# ---- the fixed-point ("Q62") CDF contract used by the HIP kernels ---------
#
# The reference accumulates the CDF strictly left to right in fp64
# (resampling.py:500-508); no parallel scan can reproduce that rounding
# sequence.  The HIP path therefore defines the CDF in exact integer
# arithmetic, which is independent of summation order:
#     q_i = rint(W_i * 2^62)            (0 when not W_i > 0)
#     C_j = sum_{i<=j} q_i              (uint64, exact)
#     T_n = ceil(su_n * 2^62)           (exact: power-of-two scaling)
#     A_n = min(#{j : C_j < T_n}, N-1)  <=> smallest j with su_n <= C_j / 2^62
# The boundary rule (strict '>' advances, ties go to the lower index) and the
# final clamp are the reference's (resampling.py:505-508).
# Any difference from the sequential-fp64 answer can only occur where su_n is
# within the fp64 round-off of the sequential CDF (``audit_near_ties``).

Q62 = float(2 ** 62)


def q62_weights(W):
    W = np.asarray(W, dtype=np.float64)
    q = np.rint(np.where(W > 0, W, 0.0) * Q62)
    return q.astype(np.uint64)


def q62_threshold(su):
    su = np.asarray(su, dtype=np.float64)
    return np.ceil(np.maximum(su, 0.0) * Q62).astype(np.uint64)


def inverse_cdf_q62(su, W):
    C = np.cumsum(q62_weights(W), dtype=np.uint64)
    T = q62_threshold(su)
    A = np.searchsorted(C, T, side="left").astype(np.int64)
    return np.minimum(A, W.shape[0] - 1)


def exp_contract(x):
    """exp(x), x <= 0, as the device forms it (oracle.c orc_exp_nonpos: Cody-Waite reduction +
    degree-13 polynomial, IEEE operations only)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    clib().orc_exp_nonpos_v(_dp(x), x.size, _dp(out))
    return out


def weights_pk(lw, K):
    """e_i = p_i 2^(k_i - K): the contract's weights relative to the reference exponent K
    (oracle.c orc_weights_pk); with K of the island and 1/s: W_i = e_i / s."""
    lw = np.ascontiguousarray(lw, dtype=np.float64)
    out = np.empty_like(lw)
    clib().orc_weights_pk(_dp(lw), lw.size, float(K), _dp(out))
    return out


def tile_partials(lw, with_q=False):
    """(K_b, S_b, SS_b) of every aligned tile of 1024 log-weights, the contract's summation
    tree (oracle.c orc_tile_partials); with_q: also the integer weights q_i = rint(e_i 2^49)."""
    lw = np.ascontiguousarray(lw, dtype=np.float64)
    nt = (lw.size + 1023) // 1024
    pK, ps, pss = np.empty(nt), np.empty(nt), np.empty(nt)
    q = np.zeros(lw.size, dtype=np.uint64) if with_q else None
    up = ctypes.POINTER(ctypes.c_uint64)
    clib().orc_tile_partials(_dp(lw), lw.size, _dp(pK), _dp(ps), _dp(pss),
                             q.ctypes.data_as(up) if with_q else None)
    return (pK, ps, pss, q) if with_q else (pK, ps, pss)


def two_level_reduce(pK, ps, pss):
    """Island level of the contract: dict(K, s, ss, ESS, rs) and the integer shares Q_b, G_b
    (integer-valued doubles below 2^53)."""
    nt = pK.size
    out = np.empty(5)
    Q, G = np.empty(nt), np.empty(nt)
    clib().orc_two_level_reduce(_dp(pK), _dp(ps), _dp(pss), nt, _dp(out), _dp(Q), _dp(G))
    return dict(K=out[0], s=out[1], ss=out[2], ESS=out[3], rs=out[4]), Q, G


def inverse_cdf_2level_c(scheme, u, lw):
    """The whole contract in C (orc_inverse_cdf_2level; counts per parent, 128-bit products):
    ancestors for `scheme` ('systematic': u = the one uniform, 'stratified': u = N uniforms,
    'multinomial': u = the N sorted uniforms) and the island reduction.  Fast enough for N = 2^22."""
    lw = np.ascontiguousarray(lw, dtype=np.float64)
    u = np.ascontiguousarray(np.atleast_1d(u), dtype=np.float64)
    A = np.empty(lw.size, dtype=np.int64)
    red = np.empty(5)
    rc = clib().orc_inverse_cdf_2level(_dp(lw), lw.size, {"multinomial": 0, "stratified": 1, "systematic": 2}[scheme],
                                       _dp(u), A.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _dp(red))
    if rc:
        raise ValueError("two-level contract: 1024 < N <= 2^30")
    return A, dict(K=red[0], s=red[1], ss=red[2], ESS=red[3], rs=red[4])


def inverse_cdf_2level(su, lw, tile=1024):
    """The two-level exact CDF of the fused step loop for N = 2^k >= 2 tiles
    (particles_amd/csrc/smc_filter_kernels.h, k_ancestors2), formulated per OFFSPRING (the C
    restatement and the kernels count per parent): tile partials and island reduction as the
    contract fixes them (tile_partials, two_level_reduce), tile shares Q_b of the 2^52 scale with
    exclusive sums G_b; inside a tile the integer CDF C_j of q_i = rint(e_i 2^49), total t_b.
    Offspring n with threshold T_n = ceil(su_n 2^52) in (G_b, G_b + Q_b] takes the first parent j
    of tile b with T_n - G_b <= floor(C_j Q_b / t_b) -- exact rational comparisons (Python
    integers here); thresholds beyond the last share go to the last particle, as
    resampling.py:500-509 would clamp."""
    lw = np.asarray(lw, dtype=np.float64)
    N = lw.shape[0]
    nt = (N + tile - 1) // tile
    assert nt >= 1 and tile == 1024
    pK, ps, pss, q = tile_partials(lw, with_q=True)
    q = np.concatenate([q, np.zeros(nt * tile - N, dtype=q.dtype)])      # a ragged last tile: zero weights
    _, Qa, Ga = two_level_reduce(pK, ps, pss)
    Q = [int(v) for v in Qa]
    G = [int(v) for v in Ga] + [int(Ga[-1]) + int(Qa[-1])]
    C = np.cumsum(q.reshape(nt, tile).astype(np.int64), axis=1)  # inclusive, exact (< 2^63)
    su = np.asarray(su, dtype=np.float64)
    T = [int(v) for v in np.ceil(np.maximum(su, 0.0) * float(2 ** 52))]
    A = np.empty(len(T), dtype=np.int64)
    b = 0
    for n, Tn in enumerate(T):                                # su sorted: one sweep over the tiles
        while b < nt - 1 and Tn > G[b + 1]:
            b += 1
        if Tn > G[nt]:
            A[n] = N - 1
            continue
        tau, tb, Qb = Tn - G[b], int(C[b, -1]), Q[b]
        lo, hi = 0, tile - 1                                  # first j with tau <= floor(C_j Q_b / t_b)
        while lo < hi:
            mid = (lo + hi) // 2
            if tb and tau <= (int(C[b, mid]) * Qb) // tb:
                hi = mid
            else:
                lo = mid + 1
        A[n] = b * tile + lo
    return A


def audit_near_ties(su, W, A_a, A_b):
    """Certify that every place two ancestor vectors differ is a near-tie.

    A mismatch at n is accepted only if su[n] lies within N*2^-52 (the worst
    case fp64 accumulation error of either CDF) of the extended-precision
    CDF at every boundary between the two answers.  Returns (n_mismatch, ok).
    """
    bad = np.nonzero(A_a != A_b)[0]
    if bad.size == 0:
        return 0, True
    cdf = np.cumsum(W.astype(np.longdouble))
    tol = np.longdouble(W.shape[0]) * np.longdouble(2.0) ** -52
    ok = True
    for n in bad:
        lo, hi = sorted((int(A_a[n]), int(A_b[n])))
        gap = np.abs(cdf[lo:hi] - np.longdouble(su[n])).max()
        ok &= bool(gap <= tol)
    return int(bad.size), ok


# --------------------------------------------------------------------------
# Distributions                   (particles/distributions.py)
# --------------------------------------------------------------------------

def normal_rvs(loc, scale, z):
    """distributions.py:270-271: ``random.normal(loc, scale, size)`` is
    ``loc + scale * standard_normal(size)`` bit for bit (numpy legacy)."""
    return loc + scale * z


def normal_logpdf(x, loc=0.0, scale=1.0):
    """distributions.py:273-274 -> scipy.stats.norm.logpdf:
    ``y=(x-loc)/scale ; -y**2/2.0 - _norm_pdf_logC - log(scale)``."""
    y = (np.asarray(x) - loc) / scale
    return -y ** 2 / 2.0 - C_NORM - np.log(scale)


def poisson_logpmf(k, mu):
    """distributions.py:528-529 -> scipy.stats.poisson.logpmf, whose ``_logpmf`` is
    ``xlogy(k, mu) - gammaln(k + 1) - mu`` behind rv_discrete's guards (mu < 0 or NaN -> NaN;
    k < 0 or not an integer -> -inf)."""
    k, mu = np.broadcast_arrays(np.asarray(k, dtype=np.float64), np.asarray(mu, dtype=np.float64))
    with np.errstate(all="ignore"):
        r = ssp_special.xlogy(k, mu) - ssp_special.gammaln(k + 1.0) - mu
    r = np.where((k < 0) | (np.floor(k) != k), -np.inf, r)
    return np.where(~(mu >= 0) | np.isnan(k), np.nan, r)


def mvnormal_rvs(loc, scale, L, z):
    """distributions.py:946-947, 961-969: ``loc + scale * dot(z, L.T)``."""
    return loc + scale * np.dot(z, L.T)


def mvnormal_logpdf(x, loc, scale, L):
    """distributions.py:949-959."""
    dim = L.shape[0]
    halflogdetcor = np.sum(np.log(np.diag(L)))
    xc = (x - loc) / scale
    z = sla.solve_triangular(L, np.transpose(xc), lower=True)
    if np.asarray(scale).ndim == 0:
        logdet = dim * np.log(scale)
    else:
        logdet = np.sum(np.log(scale), axis=-1)
    logdet += halflogdetcor
    return -0.5 * np.sum(z * z, axis=0) - logdet - dim * HALFLOG2PI


# --------------------------------------------------------------------------
# RNG plumbing (the reference uses the numpy legacy global state)
# --------------------------------------------------------------------------

class LegacyRNG:
    """numpy.random global generator, as used by resampling.py:135 and
    distributions.py:209 (``stats.norm.rvs(size=(N,d))`` draws the same stream
    as ``standard_normal((N,d))``)."""

    def rand(self, k):
        return np.random.rand(k)

    def standard_normal(self, shape):
        return np.random.standard_normal(shape)


class RecordingRNG:
    """Wraps an RNG and keeps the tape of draws, in consumption order."""

    def __init__(self, inner=None):
        self.inner = LegacyRNG() if inner is None else inner
        self.tape = []

    def rand(self, k):
        u = self.inner.rand(k)
        self.tape.append(("u", u))
        return u

    def standard_normal(self, shape):
        z = self.inner.standard_normal(shape)
        self.tape.append(("z", z))
        return z


class ReplayRNG:
    def __init__(self, tape):
        self.tape = list(tape)
        self.pos = 0

    def _next(self, kind, size):
        k, a = self.tape[self.pos]
        self.pos += 1
        assert k == kind and a.size == size, "replay tape out of sync"
        return a

    def rand(self, k):
        return self._next("u", k)

    def standard_normal(self, shape):
        return self._next("z", int(np.prod(shape))).reshape(shape)


# --------------------------------------------------------------------------
# State-space models on the hot path
# --------------------------------------------------------------------------

class LinGauss:
    """kalman.py:397-452 (``LinearGauss``); README.md:66-72 ``ToySSM`` is the
    case rho=1, sigmaX=1, sigma0=1."""
    dim = 1

    def __init__(self, rho=0.9, sigmaX=1.0, sigmaY=0.2, sigma0=None):
        self.rho, self.sigmaX, self.sigmaY = rho, sigmaX, sigmaY
        self.sigma0 = sigmaX / np.sqrt(1.0 - rho ** 2) if sigma0 is None else sigma0

    def px0(self):                       # kalman.py:427-428
        return 0.0, self.sigma0

    def px(self, xp):                    # kalman.py:430-431
        return self.rho * xp, self.sigmaX

    def py_logpdf(self, y, xp, x):       # kalman.py:433-434
        return normal_logpdf(y, loc=x, scale=self.sigmaY)

    # optimal proposal, kalman.py:436-446
    def proposal0(self, y0):
        sig2post = 1.0 / (1.0 / self.sigma0 ** 2 + 1.0 / self.sigmaY ** 2)
        mupost = sig2post * (y0 / self.sigmaY ** 2)
        return mupost, np.sqrt(sig2post)

    def proposal(self, xp, yt):
        sig2post = 1.0 / (1.0 / self.sigmaX ** 2 + 1.0 / self.sigmaY ** 2)
        mupost = sig2post * (self.rho * xp / self.sigmaX ** 2 + yt / self.sigmaY ** 2)
        return mupost, np.sqrt(sig2post)

    def logeta(self, x, y_next):         # kalman.py:448-452
        return normal_logpdf(y_next, self.rho * x, np.sqrt(self.sigmaX ** 2 + self.sigmaY ** 2))

    def kalman_matrices(self):
        return (np.atleast_2d(self.rho), np.atleast_2d(1.0),
                np.atleast_2d(self.sigmaX ** 2), np.atleast_2d(self.sigmaY ** 2),
                np.zeros(1), np.atleast_2d(self.sigma0 ** 2))


def ToySSM(sigma=0.2):
    return LinGauss(rho=1.0, sigmaX=1.0, sigmaY=sigma, sigma0=1.0)


class StochVol:
    """state_space_models.py:446-473."""
    dim = 1

    def __init__(self, mu=-1.02, rho=0.9702, sigma=0.178):
        self.mu, self.rho, self.sigma = mu, rho, sigma

    def sig0(self):                      # :458-460
        return self.sigma / np.sqrt(1.0 - self.rho ** 2)

    def px0(self):                       # :462-463
        return self.mu, self.sig0()

    def px(self, xp):                    # :465-470
        return (1.0 - self.rho) * self.mu + self.rho * xp, self.sigma

    def py_logpdf(self, y, xp, x):       # :472-473
        return normal_logpdf(y, loc=0.0, scale=np.exp(0.5 * x))

    # Pitt & Shephard's proposal and auxiliary function (:475-498)
    def _xhat(self, xst, sig, yt):
        return xst + 0.5 * sig ** 2 * (yt ** 2 * np.exp(-xst) - 1.0)

    def proposal0(self, y0):
        return self._xhat(0.0, self.sig0(), y0), self.sig0()

    def proposal(self, xp, yt):
        return self._xhat(self.px(xp)[0], self.sigma, yt), self.sigma

    def logeta(self, x, y_next):
        xst = self.px(x)[0]
        xstmmu = xst - self.mu
        xhat = self._xhat(xst, self.sigma, y_next)
        xhatmmu = xhat - self.mu
        return 0.5 / self.sigma ** 2 * (xhatmmu ** 2 - xstmmu ** 2) - 0.5 * y_next ** 2 * np.exp(-xst) * (1.0 + xstmmu)


class StochVolLeverage(StochVol):
    """state_space_models.py:501-541: PY depends on (x_{t-1}, x_t)."""

    def __init__(self, mu=-1.02, rho=0.9702, sigma=0.178, phi=0.0):
        StochVol.__init__(self, mu, rho, sigma)
        self.phi = phi

    def py_logpdf(self, y, xp, x):       # :531-541 (xp is None at t = 0)
        if xp is None:
            u = (x - self.mu) / self.sig0()
        else:
            u = (x - self.px(xp)[0]) / self.sigma
        std_x = np.exp(0.5 * x)
        return normal_logpdf(y, loc=std_x * self.phi * u,
                             scale=std_x * np.sqrt(1.0 - self.phi ** 2))


class Gordon:
    """state_space_models.py:546-577 (``Gordon_etal``); the transition depends on t."""
    dim = 1
    time_dependent = True

    def __init__(self, a=0.05, b=0.5, c=25.0, d=8.0, e=1.2, sigmaX=3.162278):
        self.a, self.b, self.c, self.d, self.e, self.sigmaX = a, b, c, d, e, sigmaX

    def px0(self):                       # :565-566
        return 0.0, 2.0

    def px(self, xp, t):                 # :568-574
        return (self.b * xp + self.c * xp / (1.0 + xp ** 2)
                + self.d * np.cos(self.e * (t - 1)), self.sigmaX)

    def py_logpdf(self, y, xp, x):       # :576-577
        return normal_logpdf(y, loc=self.a * x ** 2, scale=1.0)


class ThetaLogistic:
    """state_space_models.py:657-683."""
    dim = 1

    def __init__(self, tau0=0.15, tau1=0.12, tau2=0.1, sigmaX=0.47, sigmaY=0.39):
        self.tau0, self.tau1, self.tau2, self.sigmaX, self.sigmaY = tau0, tau1, tau2, sigmaX, sigmaY

    def px0(self):                       # :674-675
        return 0.0, 1.0

    def px(self, xp):                    # :677-680
        return xp + self.tau0 - self.tau1 * np.exp(self.tau2 * xp), self.sigmaX

    def py_logpdf(self, y, xp, x):       # :682-683
        return normal_logpdf(y, loc=x, scale=self.sigmaY)


class DiscreteCox:
    """state_space_models.py:611-630: Y_t | x ~ Poisson(exp(x))."""
    dim = 1

    def __init__(self, mu=0.0, sigma=1.0, phi=0.95):
        self.mu, self.sigma, self.phi = mu, sigma, phi

    def px0(self):                       # :621-624
        return self.mu, self.sigma / np.sqrt(1.0 - self.phi ** 2)

    def px(self, xp):                    # :626-627
        return self.mu + self.phi * (xp - self.mu), self.sigma

    def py_logpdf(self, y, xp, x):       # :629-630 -> distributions.py:528-529
        return poisson_logpmf(y, np.exp(x))


class MVLinGauss:
    """kalman.py:296-361 (``MVLinearGauss``)."""

    def __init__(self, F=None, G=None, covX=None, covY=None, mu0=None, cov0=None):
        self.covX, self.covY = np.atleast_2d(covX), np.atleast_2d(covY)
        self.dx, self.dy = self.covX.shape[0], self.covY.shape[0]
        self.mu0 = np.zeros(self.dx) if mu0 is None else mu0
        self.cov0 = self.covX if cov0 is None else np.atleast_2d(cov0)
        self.F = np.eye(self.dx) if F is None else np.atleast_2d(F)
        self.G = np.eye(self.dy, self.dx) if G is None else np.atleast_2d(G)
        self.dim = self.dx

    def kalman_matrices(self):
        return self.F, self.G, self.covX, self.covY, self.mu0, self.cov0

    def logeta(self, x, y_next):
        """kalman.py:358-361: ``logpyt`` of ``filter_step_asarray`` on the prediction (F x, covX)."""
        return kalman_filter_step(self.G, self.covY, np.matmul(x, self.F.T), self.covX, y_next)[2]


def Guarniero(alpha=0.4, dx=2):
    """kalman.py:364-394."""
    F = np.empty((dx, dx))
    for i in range(dx):
        for j in range(dx):
            F[i, j] = alpha ** (1 + abs(i - j))
    return MVLinGauss(F=F, G=np.eye(dx), covX=np.eye(dx), covY=np.eye(dx))


def _dotdot(a, b, c):
    return np.dot(np.dot(a, b), c)                       # kalman.py:157-158


def _dotdotinv(a, b, c):
    return sla.solve(c, np.dot(a, b).T, assume_a="pos", overwrite_b=True).T  # :161-163


def kalman_filter_step(G, covY, pred_mean, pred_cov, yt):
    """kalman.py:196-229 (``filter_step``): returns filt_mean, filt_cov, logpyt."""
    data_pred_mean = np.matmul(pred_mean, G.T)
    data_pred_cov = _dotdot(G, pred_cov, G.T) + covY
    if covY.shape[0] == 1:
        logpyt = normal_logpdf(yt, loc=data_pred_mean, scale=np.sqrt(data_pred_cov))
    else:
        logpyt = mvnormal_logpdf(yt, data_pred_mean, 1.0,
                                 np.linalg.cholesky(data_pred_cov))
    residual = yt - data_pred_mean
    gain = _dotdotinv(pred_cov, G.T, data_pred_cov)
    filt_mean = pred_mean + np.matmul(residual, gain.T)
    filt_cov = pred_cov - _dotdot(gain, G, pred_cov)
    return filt_mean, filt_cov, logpyt


def kalman_loglik(model, data):
    """kalman.py:483-505 (``Kalman.filter``): exact log-likelihood and
    filtering means of a linear Gaussian model -- the analytic KAT."""
    F, G, covX, covY, mu0, cov0 = model.kalman_matrices()
    ll, means = 0.0, []
    for t, yt in enumerate(data):
        if t == 0:
            pm, pc = mu0, cov0
        else:
            pm = np.matmul(fm, F.T)                       # kalman.py:169-193
            pc = _dotdot(F, fc, F.T) + covX
        fm, fc, lp = kalman_filter_step(G, covY, pm, pc, np.asarray(yt))
        ll += float(np.squeeze(lp))
        means.append(np.squeeze(fm))
    return ll, np.array(means)


# --------------------------------------------------------------------------
# The SMC step loop               (particles/core.py:299-383)
# --------------------------------------------------------------------------

class StepCtx:
    """Per-run constants of ``propagate`` (Cholesky factors of the multivariate models)."""

    def __init__(self, model, fk, y0):
        self.mv = getattr(model, "dim", 1) > 1
        if self.mv:
            self.d = model.dim
            self.LX = np.linalg.cholesky(model.covX)           # distributions.py:937
            self.LY = np.linalg.cholesky(model.covY)
            self.L0 = np.linalg.cholesky(model.cov0)
            if fk in ("guided", "apf"):
                # kalman.py:353-356 proposal0 ; filter_step with scalar-shaped mean
                self.f0m, f0c, _ = kalman_filter_step(model.G, model.covY, model.mu0,
                                                      model.cov0, np.asarray(y0))
                self.Lp0 = np.linalg.cholesky(f0c)


def propagate(model, fk, t, yt, Xp, z, ctx):
    """Move and weigh of ONE step given the (resampled) parents Xp and the standard normals z:
    core.py:315-324 with Bootstrap / GuidedPF (state_space_models.py:326-333, :374-392).
    Returns (X_t, weight increment).  ``run_filter`` iterates it; the parity tests also call it
    step by step on the device's own X_{t-1}[A_t] (teacher forcing)."""
    if fk == "apfboot":
        fk = "bootstrap"                                       # AuxiliaryBootstrap(Bootstrap): the bootstrap move and logG
    if ctx.mv:
        if fk == "apf":
            fk = "guided"                                      # AuxiliaryPF(GuidedPF): the same move and logG
        if t == 0:
            if fk == "guided":
                X = mvnormal_rvs(ctx.f0m, 1.0, ctx.Lp0, z)     # state_space_models.py:374-375
            else:
                X = mvnormal_rvs(model.mu0, 1.0, ctx.L0, z)    # kalman.py:339-340
        else:
            m = np.dot(Xp, model.F.T)                          # kalman.py:342-343
            if fk == "guided":                                 # kalman.py:348-351
                pm, pc, _ = kalman_filter_step(model.G, model.covY, m, model.covX, yt)
                Lp = np.linalg.cholesky(pc)
                X = mvnormal_rvs(pm, 1.0, Lp, z)
            else:
                X = mvnormal_rvs(m, 1.0, ctx.LX, z)
        lpy = mvnormal_logpdf(yt, np.dot(X, model.G.T), 1.0, ctx.LY)   # kalman.py:345-346
        if fk == "guided":                                     # state_space_models.py:380-392
            if t == 0:
                inc = (mvnormal_logpdf(X, model.mu0, 1.0, ctx.L0) + lpy
                       - mvnormal_logpdf(X, ctx.f0m, 1.0, ctx.Lp0))
            else:
                inc = (mvnormal_logpdf(X, m, 1.0, ctx.LX) + lpy
                       - mvnormal_logpdf(X, pm, 1.0, Lp))
        else:
            inc = lpy
        return X, inc
    if fk in ("guided", "apf"):
        if t == 0:
            loc, scale = model.proposal0(yt)
            X = normal_rvs(loc, scale, z)
            l0, s0 = model.px0()
            inc = (normal_logpdf(X, l0, s0) + model.py_logpdf(yt, None, X)
                   - normal_logpdf(X, loc, scale))
        else:
            loc, scale = model.proposal(Xp, yt)
            X = normal_rvs(loc, scale, z)
            l1, s1 = model.px(Xp)
            inc = (normal_logpdf(X, l1, s1) + model.py_logpdf(yt, Xp, X)
                   - normal_logpdf(X, loc, scale))
        return X, inc
    if t == 0:
        loc, scale = model.px0()
    else:
        loc, scale = model.px(Xp, t) if getattr(model, "time_dependent", False) else model.px(Xp)
    X = normal_rvs(loc, scale, z)
    return X, model.py_logpdf(yt, Xp if t > 0 else None, X)    # state_space_models.py:332-333


def run_filter(model, data, N, scheme="systematic", ESSrmin=0.5, fk="bootstrap",
               rng=None, keep=False, cdf="seq", T=None):
    """core.py:369-383 ``SMC.__next__`` iterated to T, for the model families
    on the hot path.  RNG consumption order per SURVEY appendix A:
    t=0: standard_normal(N[,d]);  t>=1: [if ESS < N*ESSrmin: scheme uniforms]
    then standard_normal(N[,d]).

    cdf: "seq" the reference's sequential fp64 CDF (resampling.py:500-509), "q62" / "2level"
    the device's exact integer contracts (then the resample decision also uses the contract's
    ESS, so that such a run is the device's run bit for bit).

    Returns a dict of per-step lists (ESS, log_mean, loglt, logLt, rs_flag) and
    the final X, Xp, A, lw, W; with keep=True also every step's X/A/lw/W.
    """
    rng = LegacyRNG() if rng is None else rng
    T = len(data) if T is None else T
    out = {k: [] for k in ("ESS", "log_mean", "loglt", "logLt", "rs_flag")}
    hist = {k: [] for k in ("X", "A", "lw", "W")}
    wgts = Weights()
    X = Xp = A = None
    logLt = 0.0
    log_mean_w = None
    ctx = StepCtx(model, fk, data[0])
    zshape = (N, ctx.d) if ctx.mv else N
    for t in range(T):
        yt = np.asarray(data[t])
        if fk in ("guided", "apf") and not ctx.mv:
            yt = yt.reshape(-1)[0]                        # (the laws of the proposal index data[t] as a scalar)
        rs_flag = False
        # ---- generate_particles / resample_move      core.py:315-337
        if t > 0:
            aux = wgts
            if fk in ("apf", "apfboot"):                  # core.py:307-313 setup_auxiliary_weights
                logetat = model.logeta(X, np.asarray(data[t]) if ctx.mv else np.asarray(data[t]).reshape(-1)[0])
                aux = wgts.add(logetat)
            ess = aux.ESS
            if cdf == "2level":
                ess = two_level_reduce(*tile_partials(aux.lw))[0]["ESS"]
            rs_flag = bool(ess < N * ESSrmin)             # core.py:181-183, 327
            if rs_flag:
                u = rng.rand(N_UNIFORMS[scheme](N))
                su = sorted_uniforms(scheme, N, u)
                if cdf == "seq":
                    A = inverse_cdf(su, aux.W)
                elif cdf == "2level":
                    A = inverse_cdf_2level_c(scheme, su if scheme == "multinomial" else u, aux.lw)[0]
                else:
                    A = inverse_cdf_q62(su, aux.W)
                Xp = X[A]                                 # core.py:332
                if fk in ("apf", "apfboot"):              # core.py:299-305 reset_weights
                    wgts = Weights(lw=log_mean_exp(logetat, W=wgts.W) - logetat[A])
                else:
                    wgts = Weights()
            else:
                A = np.arange(N)                          # core.py:335-336
                Xp = X
        X, inc = propagate(model, fk, t, yt, Xp, rng.standard_normal(zshape), ctx)
        # ---- reweight_particles                      core.py:323-324
        wgts = wgts.add(inc)
        # ---- compute_summaries                       core.py:351-359
        prec = log_mean_w
        log_mean_w = wgts.log_mean
        loglt = log_mean_w if (t == 0 or rs_flag) else log_mean_w - prec
        logLt += loglt
        out["ESS"].append(float(wgts.ESS))
        out["log_mean"].append(float(log_mean_w))
        out["loglt"].append(float(loglt))
        out["logLt"].append(float(logLt))
        out["rs_flag"].append(rs_flag)
        if keep:
            hist["X"].append(X.copy())
            hist["A"].append(None if A is None else A.copy())
            hist["lw"].append(wgts.lw.copy())
            hist["W"].append(wgts.W.copy())
    out.update(X=X, Xp=Xp, A=A, lw=wgts.lw, W=wgts.W, final_logLt=logLt)
    if keep:
        out["hist"] = hist
    return out


def normal_ppf(u, loc=0.0, scale=1.0):
    """distributions.py:276-277 -> scipy.stats.norm.ppf = ndtri(u) * scale + loc
    (rv_continuous.ppf: ``self._ppf(q) * scale + loc``; ``_ppf`` is special.ndtri)."""
    return ssp_special.ndtri(np.asarray(u, dtype=np.float64)) * scale + loc


def _i64(v):
    """wrap a Python int to int64 like numba / numpy scalar arithmetic does"""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >> 63 else v


def _hb_gray_decode(n):                                   # hilbert.py:206-215
    sh = 1
    while True:
        div = n >> sh
        n ^= div
        if div <= 1:
            return n
        sh <<= 1


def _hb_encode_travel(start, end, mask, i):               # hilbert.py:227-236
    travel_bit = start ^ end
    g = (i ^ (i // 2)) * (travel_bit * 2)
    return ((g | (g // (mask + 1))) & mask) ^ start


def _hb_decode_travel(start, end, mask, g):               # hilbert.py:239-244
    travel_bit = start ^ end
    modulus = mask + 1
    rg = (g ^ start) * (modulus // (travel_bit * 2))
    return _hb_gray_decode((rg | (rg // modulus)) & mask)


def hilbert_to_int(coords):
    """hilbert.py:79-91 ``Hilbert_to_int``: chunks of nD bits, one bit per coordinate (coordinate
    0 highest), most significant first (``unpack_coords`` / ``transpose_bits``, :146-190), each
    decoded through the travelling Gray code of the current sub-cube; ``pack_index`` (:134-140)
    accumulates in int64, which wraps once nD * nChunks > 63."""
    coords = [int(c) for c in coords]
    nD = len(coords)
    biggest = max(coords)
    nChunks = max(1, biggest.bit_length())                # int(ceil(log2(biggest + 1))), :149-156
    mask = 2 ** nD - 1
    start, end = 0, 2 ** ((-nChunks - 1) % nD)            # :94-99
    z = 0
    for j in range(nChunks):
        bit = nChunks - 1 - j
        chunk = 0
        for c in coords:
            chunk = chunk * 2 + ((c >> bit) & 1)
        i = _hb_decode_travel(start, end, mask, chunk)
        z = _i64((2 ** nD) * z + i)
        start_i = max(0, (i - 1) & ~1)                    # child_start_end, :287-292
        end_i = min(mask, (i + 1) | 1)
        start, end = (_hb_encode_travel(start, end, mask, start_i),
                      _hb_encode_travel(start, end, mask, end_i))
    return z


def hilbert_array(xint):                                  # hilbert.py:13-30
    return np.array([hilbert_to_int(row) for row in np.asarray(xint)], dtype=np.int64)


def hilbert_sort(x):
    """hilbert.py:33-58."""
    x = np.asarray(x)
    d = 1 if x.ndim == 1 else x.shape[1]
    if d == 1:
        return np.argsort(x, axis=0)
    xs = 1.0 / (1.0 + np.exp(-((x - np.mean(x, axis=0)) / np.std(x, axis=0))))
    maxint = np.floor(2 ** (62 / d))
    xint = np.floor(xs * maxint).astype(np.int64)
    return np.argsort(hilbert_array(xint))


def run_sqmc(model, data, N, u_tape, fk="bootstrap", cdf="seq", T=None):
    """SQMC (core.py:315-349, ``SMC(qmc=True)``) for a univariate model.  ``u_tape[t]`` are
    the points the run consumes at step t -- rqmc.sobol(N, 1) at t = 0, rqmc.sobol(N, 2)
    after -- recorded from the reference (they come from scipy's self-seeded Sobol' engine).
    Always resamples: argsort of the first coordinate, particles in sorted (= Hilbert, d = 1:
    hilbert.py:52-54) order, inverse CDF, move by the inverse-CDF transform Gamma."""
    T = len(data) if T is None else T
    out = {k: [] for k in ("ESS", "log_mean", "loglt", "logLt", "rs_flag")}
    wgts = Weights()
    X = Xp = A = None
    logLt = 0.0
    log_mean_w = None
    mv = getattr(model, "dim", 1) > 1
    if mv:                                                # as run_filter; MvNormal.ppf (:970-981) =
        LX = np.linalg.cholesky(model.covX)               # linear_transform(norm.ppf(u))
        LY = np.linalg.cholesky(model.covY)
        L0 = np.linalg.cholesky(model.cov0)
        if fk == "guided":
            f0m, f0c, _ = kalman_filter_step(model.G, model.covY, model.mu0, model.cov0,
                                             np.asarray(data[0]))
            Lp0 = np.linalg.cholesky(f0c)
    for t in range(T):
        yt = np.asarray(data[t])
        u = np.asarray(u_tape[t])
        if t == 0 and mv:
            z = normal_ppf(u)
            X = mvnormal_rvs(f0m, 1.0, Lp0, z) if fk == "guided" else mvnormal_rvs(model.mu0, 1.0, L0, z)
            rs_flag = False
        elif t == 0:                                      # core.py:315-319
            loc, scale = model.proposal0(yt) if fk == "guided" else model.px0()
            X = normal_ppf(u.squeeze(), loc, scale)       # state_space_models.py:335-336 / :394-395
            rs_flag = False
        else:                                             # core.py:339-349
            rs_flag = True
            tau = np.argsort(u[:, 0])
            h_order = hilbert_sort(X)                     # hilbert.py:33-58
            su, Ws = u[tau, 0], wgts.W[h_order]
            A = h_order[inverse_cdf(su, Ws) if cdf == "seq" else inverse_cdf_q62(su, Ws)]
            Xp = X[A]
            v = u[tau, 1:].squeeze()
            wgts = Weights()
            if mv:
                z = normal_ppf(v)
                m = np.dot(Xp, model.F.T)                 # kalman.py:342-343
                if fk == "guided":                        # kalman.py:348-351
                    pm, pc, _ = kalman_filter_step(model.G, model.covY, m, model.covX, yt)
                    Lp = np.linalg.cholesky(pc)
                    X = mvnormal_rvs(pm, 1.0, Lp, z)
                else:
                    X = mvnormal_rvs(m, 1.0, LX, z)
            else:
                if fk == "guided":
                    loc, scale = model.proposal(Xp, yt)
                else:
                    loc, scale = model.px(Xp, t) if getattr(model, "time_dependent", False) else model.px(Xp)
                X = normal_ppf(v, loc, scale)             # :338-340 / :397-398
        if mv:
            lpy = mvnormal_logpdf(yt, np.dot(X, model.G.T), 1.0, LY)  # kalman.py:345-346
            if fk == "guided":                            # state_space_models.py:380-392
                if t == 0:
                    inc = (mvnormal_logpdf(X, model.mu0, 1.0, L0) + lpy
                           - mvnormal_logpdf(X, f0m, 1.0, Lp0))
                else:
                    inc = (mvnormal_logpdf(X, m, 1.0, LX) + lpy - mvnormal_logpdf(X, pm, 1.0, Lp))
            else:
                inc = lpy
        elif fk == "guided":                              # state_space_models.py:380-392
            if t == 0:
                l0, s0 = model.px0()
                q0, qs0 = model.proposal0(yt)
                inc = (normal_logpdf(X, l0, s0) + model.py_logpdf(yt, None, X)
                       - normal_logpdf(X, q0, qs0))
            else:
                l1, s1 = model.px(Xp)
                q1, qs1 = model.proposal(Xp, yt)
                inc = (normal_logpdf(X, l1, s1) + model.py_logpdf(yt, Xp, X)
                       - normal_logpdf(X, q1, qs1))
        else:
            inc = model.py_logpdf(yt, Xp, X)
        wgts = wgts.add(inc)
        prec = log_mean_w                                 # core.py:351-359
        log_mean_w = wgts.log_mean
        loglt = log_mean_w if (t == 0 or rs_flag) else log_mean_w - prec
        logLt += loglt
        out["ESS"].append(float(wgts.ESS))
        out["log_mean"].append(float(log_mean_w))
        out["loglt"].append(float(loglt))
        out["logLt"].append(float(logLt))
        out["rs_flag"].append(rs_flag)
    out.update(X=X, Xp=Xp, A=A, lw=wgts.lw, W=wgts.W, final_logLt=logLt)
    return out


def wquantiles(W, x, alphas=(0.25, 0.50, 0.75)):
    """resampling.py:381-417 (``_wquantiles`` per column)."""
    def one(xc):
        N = W.shape[0]
        order = np.argsort(xc)
        cw = np.cumsum(W[order])
        indices = np.searchsorted(cw, alphas)
        out = []
        for a, n in zip(alphas, indices):
            prev = np.clip(n - 1, 0, N - 2)
            out.append(np.interp(a, cw[prev:prev + 2], xc[order[prev:prev + 2]]))
        return out
    if x.ndim == 1:
        return one(x)
    return np.array([one(x[:, i]) for i in range(x.shape[1])])


def compute_trajectories(A_list, N):
    """Genealogy of the final particles (smoothing.py:209-219
    ParticleHistory.compute_trajectories): B_{T-1} = arange(N), B_{t-1} = A_t[B_t];
    ``A_list[t]`` is the ancestor vector of step t (entry 0 unused)."""
    Bs = [np.arange(N)]
    for A in A_list[-1:0:-1]:
        Bs.append(A[Bs[-1]])
    Bs.reverse()
    return np.array(Bs)


# --------------------------------------------------------------------------
# Philox4x32-10 + Box-Muller: the counter-based generator the HIP path uses in
# production mode (there is no reference counterpart: the reference draws from
# MT19937).  Restated here so tests can check the device stream bit-for-bit
# for the integer part and to ~1 ulp for the Gaussian transform.
# --------------------------------------------------------------------------

PHILOX_M0, PHILOX_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
PHILOX_W0, PHILOX_W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
_M32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Salmon et al. 2011 (Random123 philox4x32, 10 rounds).  Inputs are
    broadcastable integer arrays holding 32-bit words; returns 4 uint64 arrays
    holding 32-bit words."""
    c0, c1, c2, c3, k0, k1 = [np.asarray(v).astype(np.uint64) & _M32
                              for v in np.broadcast_arrays(c0, c1, c2, c3, k0, k1)]
    for r in range(10):
        if r > 0:
            k0 = (k0 + PHILOX_W0) & _M32
            k1 = (k1 + PHILOX_W1) & _M32
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _M32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _M32
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
    return c0, c1, c2, c3


STREAM_NORMAL, STREAM_RESAMPLE, STREAM_SPACINGS = 0, 1, 2


def philox_u64_pair(seed, idx, t, island, stream):
    """The HIP path's counter layout: ctr=(idx, t, island, stream),
    key=(seed lo32, seed hi32) -> two 64-bit words (x01, x23)."""
    seed = int(seed)
    r = philox4x32_10(idx, t, island, stream, seed & 0xFFFFFFFF, seed >> 32)
    return (r[1] << np.uint64(32)) | r[0], (r[3] << np.uint64(32)) | r[2]


def u01_open(x):
    """(0,1): ((x >> 12) + 0.5) * 2^-52 (exact in fp64, never 0 or 1)."""
    return ((x >> np.uint64(12)).astype(np.float64) + 0.5) * 2.0 ** -52


def u01_halfopen(x):
    """[0,1): (x >> 11) * 2^-53 (numpy's ``rand`` convention)."""
    return (x >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def philox_normal_pair(seed, pair_idx, t, island=0, stream=STREAM_NORMAL):
    """Box-Muller on the two open-interval uniforms of one Philox call:
    (z_even, z_odd) = r*(cos, sin)(2*pi*u2), r = sqrt(-2 log u1)."""
    x01, x23 = philox_u64_pair(seed, pair_idx, t, island, stream)
    u1, u2 = u01_open(x01), u01_open(x23)
    r = np.sqrt(-2.0 * np.log(u1))
    return r * np.cos(2.0 * np.pi * u2), r * np.sin(2.0 * np.pi * u2)


def philox_normals(seed, n, t, island=0):
    """z_0..z_{n-1} of step t: particle i takes the (i&1) branch of pair i>>1."""
    p = np.arange((n + 1) // 2)
    z0, z1 = philox_normal_pair(seed, p, t, island)
    return np.stack([z0, z1], axis=1).reshape(-1)[:n]


def philox_normals_mv(seed, n, d, t, island=0):
    """(n, d) normals of step t of a multivariate filter: pair kp of particle i is Philox index
    i * ceil(d/2) + kp and yields the dimensions 2kp, 2kp + 1 (smc_filter_mv.h)."""
    hp = (d + 1) // 2
    idx = (np.arange(n, dtype=np.uint64)[:, None] * np.uint64(hp) + np.arange(hp, dtype=np.uint64)[None, :])
    z0, z1 = philox_normal_pair(seed, idx & _M32, t, island)
    return np.stack([z0, z1], axis=2).reshape(n, 2 * hp)[:, :d]


def philox_spacings(seed, M, t, island=0):
    """The device's uniform_spacings(M) in production mode (resampling.py:512-537 on the Philox
    stream): M + 1 exponential draws in fixed point, q_n = rint(-log(u_n) 2^s), s = min(57 - ceil(log2(M + 2)), 21),
    u_n the open-interval uniform of word n & 1 of Philox call n >> 1 (stream 2); su_n = Z_n / Z_M,
    Z_n = E[n >> 10] + min(q_{1024 (n >> 10)} + .. + q_n, 2^32 - 1) with E[k] the sum of the tiles of 1024 draws in
    front of tile k and Z_M = q_0 + .. + q_M (exact integers: monotone whatever the summation order; the saturation
    is what lets the device store 32-bit prefixes -- 32 standard deviations away at s = 21)."""
    lg = 0
    while (1 << lg) < M + 2:
        lg += 1
    u = philox_resample_uniforms(seed, "multinomial", M, t, island)        # M + 1 open-interval uniforms
    q = np.rint(-np.log(u) * 2.0 ** min(57 - lg, 21)).astype(np.uint64)
    tot = np.uint64(q.sum())
    nt = (M + 1 + 1023) // 1024
    qp = np.zeros(nt * 1024, dtype=np.uint64)
    qp[:M + 1] = q
    qp = qp.reshape(nt, 1024)
    E = np.concatenate([[0], np.cumsum(qp.sum(axis=1))[:-1]]).astype(np.uint64)
    z = (E[:, None] + np.minimum(np.cumsum(qp, axis=1), np.uint64(0xFFFFFFFF))).reshape(-1)[:M]
    return z.astype(np.float64) / np.float64(tot)


def philox_resample_uniforms(seed, scheme, M, t, island=0):
    """The uniforms the HIP path feeds to a scheme in production mode."""
    if scheme == "systematic":
        x01, _ = philox_u64_pair(seed, 0, t, island, STREAM_RESAMPLE)
        return u01_halfopen(np.atleast_1d(x01))
    if scheme == "stratified":
        p = np.arange((M + 1) // 2)
        x01, x23 = philox_u64_pair(seed, p, t, island, STREAM_RESAMPLE)
        return np.stack([u01_halfopen(x01), u01_halfopen(x23)], axis=1).reshape(-1)[:M]
    if scheme == "multinomial":
        p = np.arange((M + 2) // 2)
        x01, x23 = philox_u64_pair(seed, p, t, island, STREAM_SPACINGS)
        return np.stack([u01_open(x01), u01_open(x23)], axis=1).reshape(-1)[:M + 1]
    raise ValueError("%s is not a valid resampling scheme" % scheme)
