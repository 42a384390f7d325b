"""Parity checks of the HIP path against the CPU oracle, written once and run
twice: on a real MI355X by tests/test_gpu_parity.py (-m gpu, full sizes) and,
at small sizes, through the fiber emulator by tests/test_emu_parity.py so the
kernel logic is exercised in the GPU-less build container too.

Tolerances (stated per check): integer / index results are bit-exact; floating
point results that only involve IEEE + - * / are bit-exact; results that go
through exp/log/sincos (device libm differs from numpy's in the last ulp) are
compared at <= 1e-12 relative; log-evidence at <= 1e-9 relative (north star:
1e-6).
"""
import ctypes

import os

import numpy as np
import pytest

import particles_amd as pa
from oracle import smc_oracle as orc
from particles_amd import _lib
from particles_amd._lib import DeviceArray
from particles_amd import distributions as dists
from particles_amd import kalman
from particles_amd import resampling as rs
from particles_amd import state_space_models as ssm

RTOL = 1e-12


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    den = np.maximum(np.abs(b), 1e-300)
    return float(np.max(np.abs(a - b) / den)) if a.size else 0.0


# ------------------------------------------------------------------ a-4 ----
def check_weights(golden):
    g = golden("weights")
    lw = g["lw_in"].copy()
    w = rs.Weights(lw=lw)
    assert np.array_equal(lw, g["lw_after"])                 # NaN -> -inf in the caller's array
    assert rel(w.W, g["W"]) < RTOL
    assert abs(w.ESS / g["ESS"] - 1) < RTOL and abs(w.log_mean - g["log_mean"]) < 1e-12
    w2 = w.add(g["lw2"] - g["lw_after"])
    mask = np.isfinite(g["lw2"])
    assert rel(w2.W[mask], g["W2"][mask]) < 1e-10 and abs(w2.ESS / g["ESS2"] - 1) < 1e-10
    assert abs(rs.log_sum_exp(g["lw_after"]) - g["lse"]) < 1e-12
    assert abs(rs.log_mean_exp(g["lw_after"]) - g["lme"]) < 1e-12
    assert abs(rs.essl(g["lw_after"]) / g["essl"] - 1) < RTOL
    assert rel(rs.exp_and_normalise(g["lw_after"]), g["ean"]) < RTOL
    assert abs(rs.log_mean_exp(g["lw2"], W=g["W"]) - g["lme_w"]) < 1e-11
    mv = rs.wmean_and_var(g["W"], np.sin(np.arange(1000.0)))
    assert abs(mv["mean"] - g["wmean"]) < 1e-13 and abs(mv["var"] - g["wvar"]) < 1e-13


def check_wquantiles(golden, N=200001):
    """rs.wquantiles (resampling.py:381-417): device sort + running sums against the
    reference's values, and against the oracle on a larger weighted sample."""
    g = golden("weights")
    x = np.sin(np.arange(1000.0))
    q = rs.wquantiles(g["W"], x, alphas=(0.05, 0.25, 0.5, 0.75, 0.999))
    assert np.allclose(q, g["wq"], rtol=1e-12, atol=1e-13)
    x2 = np.stack([x, np.cos(3.0 * np.arange(1000.0))], axis=1)
    q2 = rs.wquantiles(g["W"], x2)
    assert q2.shape == (2, 3) and np.allclose(q2, g["wq2"], rtol=1e-12, atol=1e-13)
    rng = np.random.default_rng(8)
    W = orc.exp_and_normalise(1.5 * rng.standard_normal(N))
    xs = rng.standard_normal(N)
    al = (0.001, 0.1, 0.5, 0.9, 0.9999)
    assert np.allclose(rs.wquantiles(W, xs, alphas=al), orc.wquantiles(W, xs, alphas=al),
                       rtol=1e-9, atol=1e-9)


def check_wmean_and_cov(golden, N=150001):
    """rs.wmean_and_cov / wmean_and_var_str_array / wquantiles_str_array (resampling.py:341-380, 420-442) on the
    device against the reference's own outputs (fixture moments_cov) and, on a larger sample and at d = 1 .. 32,
    against the oracle's restatement."""
    g = golden("moments_cov")
    m, c = rs.wmean_and_cov(g["W"], g["X"])
    assert np.allclose(m, g["mean5"], rtol=1e-13, atol=1e-14) and np.allclose(c, g["cov5"], rtol=1e-12, atol=1e-13)
    assert c.shape == (5, 5) and np.array_equal(c, c.T)
    m1, c1 = rs.wmean_and_cov(g["W"], g["X"][:, 0])
    assert abs(m1 - g["mean1"]) < 1e-13 and np.ndim(c1) == 0 and abs(float(c1) - float(g["cov1"])) < 1e-12
    xs = np.zeros(len(g["W"]), dtype=[("a", float), ("b", float)])
    xs["a"], xs["b"] = g["sa"], g["sb"]
    mv = rs.wmean_and_var_str_array(g["W"], xs)
    assert abs(mv["mean"]["a"] - g["sm_a"]) < 1e-13 and abs(mv["var"]["b"] - g["sv_b"]) < 1e-12
    wq = rs.wquantiles_str_array(g["W"], xs, alphas=(0.1, 0.5, 0.9))
    assert np.allclose(wq["a"], g["sq_a"], rtol=1e-12, atol=1e-13) and np.allclose(wq["b"], g["sq_b"], rtol=1e-12, atol=1e-13)
    rng = np.random.default_rng(4)
    W = orc.exp_and_normalise(1.5 * rng.standard_normal(N))
    for d in (1, 2, 7, 32, 40):                       # (40: beyond the kernel's tile -- the host route)
        X = rng.standard_normal((N, d)) @ rng.standard_normal((d, d)) + 3.0
        m, c = rs.wmean_and_cov(W, X)
        mo, co = orc.wmean_and_cov(W, X)
        assert np.allclose(m, mo, rtol=1e-11, atol=1e-12) and np.allclose(c, np.atleast_2d(co), rtol=1e-9, atol=1e-10), d


def check_device_sort(sizes=(1, 63, 64, 2047, 2048, 2049, 50001)):
    """The hand-written LSD radix sort (csrc/smc_sort.hip): argsort of doubles against
    np.argsort(kind="stable") -- ties, signed zeros, infinities, tiny / huge magnitudes -- and the
    Hilbert sort's signed int64 keys; a permutation, stable, for sizes around the tile edges."""
    from particles_amd import hilbert
    rng = np.random.default_rng(5)
    for N in sizes:
        x = rng.standard_normal(N) * 10.0 ** rng.integers(-300, 300, size=N)
        if N > 8:
            x[rng.integers(0, N, size=N // 3)] = x[1]                     # many ties
            x[2], x[3], x[4], x[5] = 0.0, -0.0, np.inf, -np.inf
        o = np.asarray(hilbert.argsort(x))
        assert o.dtype == np.int64 and np.array_equal(np.sort(o), np.arange(N))
        assert np.all(np.diff(x[o]) >= 0)
        ref = np.argsort(x, kind="stable")
        same_value = x[o] == x[ref]                   # (+0 and -0 compare equal: either order is a valid sort)
        assert np.all(same_value)
        nz = x[o] != 0.0
        assert np.array_equal(o[nz], ref[nz])                                 # stable: ties in index order
    pts = rng.standard_normal((4097, 2))
    order = np.asarray(hilbert.hilbert_sort(pts))
    assert np.array_equal(np.sort(order), np.arange(4097))


def check_sort_window(sizes=(8193, 20001, 50001), window_min=8193):
    """The radix sort's four-pass form (the 32 bits below the keys' highest varying bit, then k_rs_fix on the low bits;
    csrc/smc_sort.hip) gives np.argsort(kind="stable")'s permutation for: wide and narrow ranges (a state far from 0
    relative to its spread: the window must start at the highest VARYING bit), heavy duplication (runs of equal keys of
    any length need no fix-up), values a few ulps apart (groups the fix-up orders), clusters of MORE than 32 distinct keys
    inside one window value (the one-workgroup eight-pass fallback), all keys equal, two distinct keys, signed zeros and
    infinities, int64 Hilbert keys -- and the eight-pass form gives the same."""
    from particles_amd import hilbert
    rng = np.random.default_rng(11)

    def cases(N):
        yield "normal", rng.standard_normal(N)
        yield "wide", rng.standard_normal(N) * 10.0 ** rng.integers(-300, 300, size=N)
        yield "narrow", 1000.0 + 1e-3 * rng.standard_normal(N)
        yield "duplicates", rng.integers(0, 37, size=N).astype(np.float64)
        yield "all equal", np.full(N, 3.25)
        x = np.full(N, 1.0)
        x[N // 2] = np.nextafter(1.0, 2.0)
        yield "two values", x
        base = rng.standard_normal(N)
        x = base.copy()                                   # pairs / triples a few ulps apart: groups for the fix-up
        idx = rng.integers(0, N, size=N // 4)
        x[idx] = np.nextafter(base[(idx + 1) % N], np.inf)
        idx = rng.integers(0, N, size=N // 8)
        x[idx] = np.nextafter(np.nextafter(base[(idx + 2) % N], -np.inf), -np.inf)
        yield "ulps apart", x
        x = rng.standard_normal(N)                         # a cluster of 500 distinct values inside one window value
        c = 0.5 + np.arange(500) * 2.0 ** -52
        x[rng.choice(N, 500, replace=False)] = rng.permutation(c)
        yield "tight cluster", x
        par = rng.standard_normal(N // 20 + 1)              # offspring of a parent, jitter far below the window's resolution:
        cnt = rng.integers(1, 60, size=par.size)            # groups of up to 60 keys the fix-up orders itself
        x = np.repeat(par, cnt)[:N]
        x = np.concatenate([x, rng.standard_normal(N - x.size)]) if x.size < N else x
        x = rng.permutation(x + 1e-13 * rng.standard_normal(N))
        yield "jittered offspring", x
        x = rng.standard_normal(N)
        x[:6] = 0.0, -0.0, np.inf, -np.inf, 5e-324, -5e-324
        yield "specials", x
        if N != sizes[0]:
            return                                         # (the shapes below at the first size only: CPU-suite time)
        x = rng.standard_normal(N)                         # outliers, heavy tails, non-finite keys, the whole exponent range
        x[N // 3] = 3000.0
        yield "outlier 3e3", x
        x = rng.standard_normal(N)
        x[N // 3], x[N // 5] = 1e9, -1e12
        yield "outliers 1e9", x
        yield "cauchy", rng.standard_cauchy(N)
        yield "lognormal 20", np.exp(20.0 * rng.standard_normal(N))
        x = rng.standard_normal(N)
        x[rng.random(N) < 0.6] = np.nan                    # (more than half the sampled keys non-finite)
        x[::97] = np.inf
        yield "mostly nan", x
        yield "huge range", rng.standard_normal(N) * 1e308

    pts = rng.standard_normal((sizes[0], 3))
    try:
        first = None
        for wm, lm in ((window_min, 0), (1 << 40, 0)):    # four passes + fix-up from window_min keys on; eight passes
            _lib.check(_lib.lib().smc_debug_sort_window_min(wm))
            for N in sizes:
                for name, x in cases(N):
                    o = np.asarray(hilbert.argsort(x))
                    ref = np.argsort(x, kind="stable")
                    assert np.array_equal(np.sort(o), np.arange(N)), (wm, lm, N, name)
                    assert np.array_equal(x[o], x[ref], equal_nan=True), (wm, lm, N, name)
                    nz = x[o] != 0.0                          # (+0 and -0 compare equal: either order is a valid sort)
                    assert np.array_equal(o[nz], ref[nz]), (wm, lm, N, name, int(np.sum(o != ref)))
            order = np.asarray(hilbert.hilbert_sort(pts))
            if first is None:
                first = order
            else:
                assert np.array_equal(first, order)
    finally:
        _lib.check(_lib.lib().smc_debug_sort_window_min(8193))


def check_weights_edges(N):
    w = rs.Weights(lw=np.full(N, -np.inf))
    assert np.isnan(w.W).all() and np.isnan(w.ESS) and np.isnan(w.log_mean)
    e = rs.Weights()
    assert e.N == 0 and not hasattr(e, "W")
    w1 = e.add(np.zeros(N))
    assert np.array_equal(w1.W, np.full(N, 1.0 / N)) and abs(w1.ESS - N) < 1e-9 * N
    rng = np.random.default_rng(0)
    lw = 50.0 * rng.standard_normal(N)
    lw[::7] = -np.inf
    o = orc.Weights(lw=lw.copy())
    d = rs.Weights(lw=lw.copy())
    assert rel(d.W, o.W) < RTOL and abs(d.ESS / o.ESS - 1) < RTOL
    assert abs(d.log_mean - o.log_mean) < 1e-12 * max(1, abs(o.log_mean))


# ------------------------------------------------------------ a-5 / a-6 ----
def check_inverse_cdf(N, M, seed=0):
    rng = np.random.default_rng(seed)
    W = orc.exp_and_normalise(3.0 * rng.standard_normal(N))
    W[rng.integers(0, N, size=max(1, N // 50))] = 0.0
    W /= W.sum()
    su = np.sort(rng.random(M))
    A = rs.inverse_cdf(su, W)
    assert A.dtype == np.int64 and A.shape == (M,)
    assert np.array_equal(A, orc.inverse_cdf_q62(su, W))     # bit-exact vs the Q62 oracle
    try:
        A_seq = orc.inverse_cdf(su, W)
    except IndexError:
        return
    n, ok = orc.audit_near_ties(su, W, A_seq, A)             # vs the reference ordering
    assert ok and n <= max(2, M // 100000)


def check_inverse_cdf_dyadic(N, M):
    """Exactly summable weights: must equal the reference's sequential CDF 100%."""
    rng = np.random.default_rng(3)
    k = rng.integers(0, 2 ** 30 // N, size=N).astype(np.int64)
    k[-1] += 2 ** 30 - k.sum()
    W = k / 2.0 ** 30
    assert W.sum() == 1.0
    for scheme in ("systematic", "stratified"):
        u = rng.random(orc.N_UNIFORMS[scheme](M))
        su = orc.sorted_uniforms(scheme, M, u)
        A = rs.inverse_cdf(su, W)
        assert np.array_equal(A, orc.inverse_cdf(su, W))         # the reference's loop (resampling.py:500-509)
        assert np.array_equal(A, orc.inverse_cdf_q62(su, W))     # ... and the device's contract, restated


def check_schemes_vs_reference(golden):
    """Same numpy seed as the reference run -> same ancestors (near-ties audited)."""
    g = golden("resampling")
    for scheme in ("systematic", "stratified", "multinomial"):
        for M in (1500, 400, 4000):
            np.random.seed(11)
            A = rs.resampling(scheme, g["W"], M=M)
            ref = g["A_%s_%d" % (scheme, M)]
            assert A.dtype == np.int64 and A.shape == ref.shape
            if not np.array_equal(A, ref):
                np.random.seed(11)
                su = orc.sorted_uniforms(scheme, M, np.random.rand(orc.N_UNIFORMS[scheme](M)))
                n, ok = orc.audit_near_ties(su, g["W"], ref, A)
                assert ok and n <= 1


def check_residual_killing(golden):
    """residual / killing (resampling.py:611-626, 680-697) on the device: same numpy
    seed as the reference run; bit-exact against the oracle on the Q62 CDF, and equal to
    the reference's ancestors except at audited near-ties of the multinomial part."""
    g = golden("resampling2")
    W = g["W"]
    for M in (1500, 400, 4000):
        np.random.seed(11)
        A = rs.resampling("residual", W, M=M)
        np.random.seed(11)
        want = orc.residual(W, M, cdf="q62")
        ref = g["A_residual_%d" % M]
        assert A.dtype == np.int64 and A.shape == (M,) and np.array_equal(A, want)
        sip = int(np.floor(M * W).sum())
        assert np.array_equal(A[:sip], ref[:sip])             # the deterministic copies
        assert np.mean(A == ref) >= 0.999
    np.random.seed(11)
    A = rs.resampling("killing", W, M=1500)
    np.random.seed(11)
    assert np.array_equal(A, orc.killing(W, 1500, cdf="q62"))
    assert np.mean(A == g["A_killing_1500"]) >= 0.999
    import pytest
    with pytest.raises(ValueError, match="killing resampling defined only for M=N"):
        rs.resampling("killing", W, M=100)
    # M W integral: no residual draw, and no uniform is consumed (the reference draws none)
    np.random.seed(11)
    A = rs.resampling("residual", g["W_integral"], M=64)
    assert np.array_equal(A, g["A_residual_integral"])
    assert np.random.rand() == np.random.RandomState(11).rand()
    # ssp (resampling.py:628-678): IEEE operations only, same uniforms -> the reference's ancestors
    for M in (1500, 400, 4000):
        np.random.seed(11)
        A = rs.resampling("ssp", W, M=M)
        assert A.dtype == np.int64 and np.array_equal(A, g["A_ssp_%d" % M])
    assert np.array_equal(rs.ssp(np.array([1.0]), 5), np.zeros(5, dtype=np.int64))
    for N in (2, 3, 7, 513, 514, 1025):          # chunk borders of the device walk, M != N
        r2 = np.random.default_rng(N)
        Wn = r2.random(N)
        Wn /= Wn.sum()
        for M in (N, 2 * N + 1):
            np.random.seed(4)
            got = rs.ssp(Wn, M)
            np.random.seed(4)
            assert np.array_equal(got, orc.ssp(Wn, M)), (N, M)
    # the whole registry of the reference
    assert set(rs.rs_funcs) == {"multinomial", "stratified", "systematic", "residual", "ssp", "killing"}
    # large, Philox draws on the device: offspring counts of residual are floor(M W) or more
    rng = np.random.default_rng(5)
    N = 50000
    Wl = orc.exp_and_normalise(2.0 * rng.standard_normal(N))
    rs.set_rng("philox")
    try:
        Ad = rs.resampling("residual", pa.DeviceArray.from_numpy(Wl), M=N).get()
        Ak = rs.resampling("killing", pa.DeviceArray.from_numpy(Wl), M=N).get()
        As = rs.resampling("ssp", pa.DeviceArray.from_numpy(Wl[:5000] / Wl[:5000].sum()), M=5000).get()
    finally:
        rs.set_rng("numpy")
    cnt = np.bincount(Ad, minlength=N)
    assert cnt.sum() == N and np.all(cnt >= np.floor(N * Wl))
    cs = np.bincount(As, minlength=5000)
    fl = np.floor(5000 * Wl[:5000] / Wl[:5000].sum())
    assert cs.sum() == 5000 and np.all(cs >= fl) and np.all(cs <= fl + 1)     # k or k+1 offspring
    kept = Ak == np.arange(N)
    assert abs(kept.mean() - np.mean(Wl / Wl.max())) < 0.02          # P(keep i) = W_i / max W
    assert Ak.min() >= 0 and Ak.max() < N


def check_schemes_replay(N, M, seed=1):
    rng = np.random.default_rng(seed)
    W = orc.exp_and_normalise(2.5 * rng.standard_normal(N))
    Wd = pa.DeviceArray.from_numpy(W)
    for scheme in ("systematic", "stratified", "multinomial"):
        u = rng.random(orc.N_UNIFORMS[scheme](M))
        su = orc.sorted_uniforms(scheme, M, u)
        want = orc.inverse_cdf_q62(su, W)
        ud = pa.DeviceArray.from_numpy(su if scheme == "multinomial" else u)
        A = pa.DeviceArray((M,), np.int64)
        _lib.check(_lib.lib().smc_resample(Wd.ctx.h, _lib.SCHEMES[scheme], Wd.ptr, N, M, ud.ptr,
                                           0, A.ptr))
        got = A.get()
        assert np.array_equal(got, want), scheme
        assert np.all(np.diff(got) >= 0) and got.min() >= 0 and got.max() < N


def check_schemes_philox(N, M, seed=77):
    rng = np.random.default_rng(seed)
    W = orc.exp_and_normalise(2.0 * rng.standard_normal(N))
    Wd = pa.DeviceArray.from_numpy(W)
    pa.seed(seed)
    for scheme in ("systematic", "stratified"):
        for counter in (5, (3 << 32) | 9):
            A = pa.DeviceArray((M,), np.int64)
            _lib.check(_lib.lib().smc_resample(Wd.ctx.h, _lib.SCHEMES[scheme], Wd.ptr, N, M, None,
                                               counter, A.ptr))
            u = orc.philox_resample_uniforms(seed, scheme, M, counter & 0xFFFFFFFF, counter >> 32)
            want = orc.inverse_cdf_q62(orc.sorted_uniforms(scheme, M, u), W)
            assert np.array_equal(A.get(), want), scheme
    # multinomial: sorted uniforms by exponential spacings, drawn on the device
    su = pa.DeviceArray((M,))
    _lib.check(_lib.lib().smc_uniform_spacings(su.ctx.h, M, 12, su.ptr))
    s = su.get()
    assert np.all(np.diff(s) >= 0) and s[0] > 0 and s[-1] < 1
    want = orc.uniform_spacings_from(orc.philox_resample_uniforms(seed, "multinomial", M, 12))
    # (fixed point: a spacing is rounded to 2^-21 of the mean spacing 1 / M, the roundings add up like a random walk)
    assert np.max(np.abs(s - want)) < 8.0 * np.sqrt(M) * 2.0 ** -21 / M + 1e-12
    assert np.max(np.abs(s - orc.philox_spacings(seed, M, 12))) < 1e-13
    A = pa.DeviceArray((M,), np.int64)
    _lib.check(_lib.lib().smc_resample(Wd.ctx.h, _lib.MULTINOMIAL, Wd.ptr, N, M, None, 12, A.ptr))
    assert np.array_equal(A.get(), orc.inverse_cdf_q62(s, W))


def check_resampling_statistics(N, reps):
    """book/resampling/compare_tv_distance_resampling.py: E[counts] = M*W and,
    for systematic, offspring in {floor, ceil}(N*W)."""
    rng = np.random.default_rng(5)
    W = orc.exp_and_normalise(rng.standard_normal(N))
    rs.set_rng("philox")
    try:
        pa.seed(99)
        Wd = pa.DeviceArray.from_numpy(W)
        for scheme in ("systematic", "stratified", "multinomial"):
            tot = np.zeros(N)
            for _ in range(reps):
                c = np.bincount(rs.resampling(scheme, Wd).get(), minlength=N)
                if scheme == "systematic":
                    assert np.all((c >= np.floor(N * W)) & (c <= np.ceil(N * W)))
                tot += c
            err = np.abs(tot / reps - N * W)
            sd = np.sqrt(N * W * (1 - W) / reps) + 1e-12
            assert np.all(err < 6 * sd + 1e-9), scheme
    finally:
        rs.set_rng("numpy")


def check_unknown_scheme():
    import pytest
    with pytest.raises(ValueError, match="not a valid resampling scheme"):
        rs.resampling("bogus", np.ones(4) / 4)
    with pytest.raises(ValueError, match="not a valid resampling scheme"):
        pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(), data=[np.zeros(1)]), N=10, resampling="bogus")


# ------------------------------------------------------------------ a-7 ----
def check_gather(N, d):
    rng = np.random.default_rng(2)
    X = rng.standard_normal((N, d)) if d > 1 else rng.standard_normal(N)
    A = rng.integers(0, N, size=N)
    Xd, Ad = pa.DeviceArray.from_numpy(X), pa.DeviceArray.from_numpy(A)
    out = pa.DeviceArray(X.shape)
    _lib.check(_lib.lib().smc_gather(Xd.ctx.h, Xd.ptr, Ad.ptr, N, d, out.ptr))
    assert np.array_equal(out.get(), X[A])


# ------------------------------------------------------------ a-2 / a-3 ----
def stats_norm_logpdf(x, loc, scale):
    """scipy.stats.norm.logpdf's expression (distributions.py:273-274)."""
    z = (np.asarray(x) - loc) / scale
    return -0.5 * z * z - np.log(scale) - 0.5 * np.log(2.0 * np.pi)


def check_normal(golden):
    g = golden("dists")
    lp = dists.Normal(loc=g["loc"], scale=0.7).logpdf(g["x"])
    assert np.max(np.abs(lp - g["normal_logpdf"])) < 1e-15 * 8
    lp = dists.Normal(loc=0.0, scale=np.exp(0.5 * g["x"])).logpdf(np.array([0.3]))
    assert np.max(np.abs(lp - g["normal_logpdf_sv"])) < 1e-14
    np.random.seed(10)
    z = np.random.standard_normal(64)
    x = dists.Normal(loc=g["loc"], scale=0.7).rvs(size=64, z=z)
    assert np.array_equal(x, g["normal_rvs"])                # loc + scale*z: bit-exact
    # scalar parameters live in a cache of one-element device arrays (DeviceArray.scalar: uploaded once, through the
    # kernel arguments of a one-wave launch -- smc_memcpy_h2d's small-copy form): every kind of scalar, the same values
    # again, a NaN, more distinct values than the cache holds, and a (1,) array beside an (N,) one
    from particles_amd._lib import DeviceArray as DA, _SCALARS
    xs = g["x"]
    for sc in (0.7, np.float64(0.7), 1, np.float32(0.5), np.array([0.7]), np.array(0.7)):
        for rep in range(2):
            lp = dists.Normal(loc=0.25, scale=sc).logpdf(xs)
            ref = stats_norm_logpdf(xs, 0.25, float(np.asarray(sc).reshape(-1)[0]))
            assert np.max(np.abs(lp - ref)) < 1e-14 * 8, (sc, rep)
    assert np.all(np.isnan(dists.Normal(loc=0.0, scale=float("nan")).logpdf(xs)))
    a, b = DA.scalar(0.7), DA.scalar(np.float64(0.7))
    assert a is b and a.get()[0] == 0.7 and np.isnan(DA.scalar(float("nan")).get()[0])
    for k in range(4200):                                        # (beyond the cache's 4096 entries: it starts afresh)
        DA.scalar(1.0 + k * 2.0 ** -20)
    assert len(_SCALARS) <= 4096 and DA.scalar(1.0 + 4199 * 2.0 ** -20).get()[0] == 1.0 + 4199 * 2.0 ** -20
    r = np.arange(33, dtype=np.float64)                          # (the kernel-argument copy takes up to 32 words)
    for n in (1, 2, 31, 32, 33):
        assert np.array_equal(DA.from_numpy(r[:n]).get(), r[:n]), n


def check_poisson(golden):
    """Poisson.logpdf (distributions.py:528-529) against the oracle's restatement of scipy's
    expression, guards included; then the DiscreteCox model on the generic path (PY returning
    dists.Poisson, device ops) against the same model on the fused path."""
    rng = np.random.default_rng(5)
    x = rng.normal(0.5, 1.0, 3001)
    for k in (0.0, 1.0, 7.0, 40.0):
        lp = dists.Poisson(rate=np.exp(x)).logpdf(k)
        ref = orc.poisson_logpmf(k, np.exp(x))
        assert np.max(np.abs(lp - ref) / (1.0 + np.abs(ref))) < 1e-14
    k = np.array([0.0, 3.0, -1.0, 2.5, 0.0, 4.0, 2.0, np.nan])
    rate = np.array([0.0, 0.0, 1.0, 1.0, np.inf, np.nan, -1.0, 1.0])
    lp = dists.Poisson(rate=rate).logpdf(k)
    ref = orc.poisson_logpmf(k, rate)
    assert np.array_equal(np.isnan(lp), np.isnan(ref))
    assert np.array_equal(lp[~np.isnan(ref)], ref[~np.isnan(ref)])
    assert lp[0] == 0.0 and lp[1] == -np.inf and lp[2] == -np.inf and lp[3] == -np.inf

    g = golden("cox_boot")
    y = list(g["y"])
    model = ssm.DiscreteCox(mu=0.5, sigma=0.4, phi=0.9)

    class GenericBootstrap(ssm.Bootstrap):
        def _device_model(self):
            return None

    pa.seed(11)
    fused = pa.SMC(fk=ssm.Bootstrap(ssm=model, data=y), N=8000)
    fused.run()
    pa.seed(12)
    gen = pa.SMC(fk=GenericBootstrap(ssm=model, data=y), N=8000)
    gen.run()
    assert np.isfinite(fused.logLt) and abs(fused.logLt - gen.logLt) < 0.6
    # counts outside the support: every increment is -inf, as rv_discrete.logpmf returns
    bad = list(y)
    bad[3] = np.array([2.5])
    pf = pa.SMC(fk=ssm.Bootstrap(ssm=model, data=bad), N=64)
    for _ in range(4):
        next(pf)
    assert np.all(np.isneginf(pf.wgts.lw))


def check_normal_philox(n, seed=4242):
    pa.seed(seed)
    z = pa.DeviceArray((n,))
    _lib.check(_lib.lib().smc_standard_normal(z.ctx.h, 7, n, z.ptr))
    want = orc.philox_normals(seed, n, 7)
    assert np.max(np.abs(z.get() - want)) < 1e-13
    u = pa.DeviceArray((n,))
    _lib.check(_lib.lib().smc_uniform(u.ctx.h, (2 << 32) | 7, n, u.ptr))
    p = np.arange((n + 1) // 2)
    x01, x23 = orc.philox_u64_pair(seed, p, 7, 2, orc.STREAM_RESAMPLE)
    wantu = np.stack([orc.u01_halfopen(x01), orc.u01_halfopen(x23)], axis=1).reshape(-1)[:n]
    assert np.array_equal(u.get(), wantu)                    # integer pipeline: bit-exact


# ------------------------------------------------------------------ a-8 ----
def check_mvn(golden):
    g = golden("dists")
    mv = dists.MvNormal(loc=g["mloc"], scale=1.3, cov=g["cov"])
    assert np.max(np.abs(mv.logpdf(g["x5"]) - g["mv_logpdf"])) < 1e-12
    np.random.seed(12)
    z = np.random.standard_normal((40, 5))
    assert np.max(np.abs(mv.rvs(size=40, z=z) - g["mv_rvs"])) < 1e-13


def check_mvn_large(N, d):
    rng = np.random.default_rng(8)
    Amat = rng.standard_normal((d, d))
    cov = Amat @ Amat.T + d * np.eye(d)
    loc = rng.standard_normal((N, d))
    x = rng.standard_normal((N, d))
    L = np.linalg.cholesky(cov)
    mv = dists.MvNormal(loc=loc, cov=cov)
    assert rel(mv.logpdf(x), orc.mvnormal_logpdf(x, loc, 1.0, L)) < 1e-11
    z = rng.standard_normal((N, d))
    assert np.max(np.abs(mv.rvs(size=N, z=z) - orc.mvnormal_rvs(loc, 1.0, L, z))) < 1e-12
    # Philox draws: right covariance
    pa.seed(5)
    s = dists.MvNormal(loc=np.zeros(d), cov=cov).rvs(size=N)
    if N >= 20000:
        assert np.max(np.abs(np.cov(s.T) - cov)) < 0.1 * np.abs(cov).max()


def dense_mv_matrices(dx, dy, seed=11):
    """A MVLinearGauss (kalman.py:296-361) with full covariance matrices and a full observation matrix."""
    r = np.random.RandomState(seed)
    spd = lambda n, s: (lambda a: a @ a.T / n + s * np.eye(n))(r.standard_normal((n, n)))
    F = 0.4 * r.standard_normal((dx, dx)) / np.sqrt(dx) + 0.3 * np.eye(dx)
    G = r.standard_normal((dy, dx)) / np.sqrt(dx) + np.eye(dy, dx)
    return dict(F=F, G=G, covX=spd(dx, 0.5), covY=spd(dy, 0.7), mu0=0.1 * r.standard_normal(dx), cov0=spd(dx, 1.0))


# ------------------------------------------------------------------ a-1 ----
MODELS = {
    "toy": (lambda: kalman.ToySSM(0.2), lambda: orc.ToySSM(0.2)),
    "sv": (lambda: ssm.StochVol(), lambda: orc.StochVol()),
    "lg_adaptive": (lambda: kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=1.5),
                    lambda: orc.LinGauss(rho=0.9, sigmaX=1.0, sigmaY=1.5)),
    "lg_guided": (lambda: kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=0.2),
                  lambda: orc.LinGauss(rho=0.9, sigmaX=1.0, sigmaY=0.2)),
    "mv4": (lambda: kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=4),
            lambda: orc.Guarniero(alpha=0.4, dx=4)),
    "mv32": (lambda: kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=32),
             lambda: orc.Guarniero(alpha=0.4, dx=32)),
    # correlated noises, a rectangular G: none of the step's factors is diagonal (the dense MFMA products of k_propagate_mv;
    # the Guarniero models above have G = covX = covY = I and take its element-wise form, smc_filter_mv.h "DG")
    "mvd8": (lambda: kalman.MVLinearGauss(**dense_mv_matrices(8, 6)), lambda: orc.MVLinGauss(**dense_mv_matrices(8, 6))),
    "mvd32": (lambda: kalman.MVLinearGauss(**dense_mv_matrices(32, 32)), lambda: orc.MVLinGauss(**dense_mv_matrices(32, 32))),
    "gordon": (lambda: ssm.Gordon_etal(), lambda: orc.Gordon()),
    "theta": (lambda: ssm.ThetaLogistic(), lambda: orc.ThetaLogistic()),
    "svlev": (lambda: ssm.StochVolLeverage(phi=-0.5), lambda: orc.StochVolLeverage(phi=-0.5)),
    "cox": (lambda: ssm.DiscreteCox(mu=0.5, sigma=0.4, phi=0.9),
            lambda: orc.DiscreteCox(mu=0.5, sigma=0.4, phi=0.9)),
}


def tapes_from_oracle(tape, T, N, scheme):
    """Dense device tapes (T,1,N) / (T,1,K) from the oracle's consumption-ordered
    tape; multinomial slots hold the sorted uniforms (resampling.py:536-537)."""
    K = 1 if scheme == "systematic" else N
    d = tape[0][1].size // N
    z = np.zeros((T, 1, N) if d == 1 else (T, 1, N, d))
    u = np.zeros((T, 1, K))
    t = -1
    pend = None
    for kind, a in tape:
        if kind == "u":
            pend = a
        else:
            t += 1
            z[t, 0] = a.reshape(z.shape[2:])
            if pend is not None:
                u[t, 0] = orc.uniform_spacings_from(pend) if scheme == "multinomial" else pend
                pend = None
    assert t == T - 1
    return z, u


def describe(pf):
    buf = ctypes.create_string_buffer(256)
    _lib.check(_lib.lib().smc_filter_describe(pf._f, buf, 256))
    return buf.value.decode()


def audit_history(pf, mk_orc, fk, y, scheme, ESSrmin, z=None, u=None, exact=True, island=0,
                  tol=1e-12, steps=None):
    """Teacher-forced audit of a store_history=True fused run: EVERY step, EVERY entry.

    For each step t the oracle is handed the device's own state of step t-1 and must reproduce
    step t: the resample decision, the ancestors -- bit for bit against the device's exact CDF
    contract (two-level: oracle.c orc_inverse_cdf_2level from the log-weights; flat Q62:
    inverse_cdf_q62 from the weights), and against the reference's sequential fp64 CDF
    (resampling.py:500-509) with every mismatch certified a near-tie -- then X_t from
    X_{t-1}[A_t] and the step's normals, the log-weights, ESS and the evidence.  One flipped
    ancestor therefore fails the step it occurs in instead of silently ending the comparison.
    z, u: the replay tapes; None = production mode (the oracle's Philox restatement).
    Returns the number of audited near-ties against the reference CDF."""
    N, T = pf.N, pf._n
    model = mk_orc()
    ctx = orc.StepCtx(model, fk, y[0])
    two_level = "k_ancestors2" in describe(pf)
    summ = pf._summ()[island]
    near_ties = 0
    logLt = 0.0
    prev_log_mean = None
    lw_prev = X_prev = None
    for t in (range(T) if steps is None else steps):
        X = pf._history(_lib.FIELD_X, t, island)
        lw = pf._history(_lib.FIELD_LW, t, island)
        if t > 0 and (lw_prev is None):
            X_prev = pf._history(_lib.FIELD_X, t - 1, island)
            lw_prev = pf._history(_lib.FIELD_LW, t - 1, island)
        rs_flag = bool(summ[t, 4])
        Xp = None
        if t > 0:
            # ---- the decision (core.py:181-183) on the contract's ESS
            if two_level:
                ess_prev = orc.two_level_reduce(*orc.tile_partials(lw_prev))[0]["ESS"]
                assert ess_prev == summ[t - 1, 0], (t, ess_prev, summ[t - 1, 0])
            else:
                ess_prev = summ[t - 1, 0]
                assert abs(ess_prev / orc.Weights(lw=lw_prev.copy()).ESS - 1) < 1e-11
            assert rs_flag == bool(ess_prev < N * ESSrmin), t
            if rs_flag:
                A = pf._history(_lib.FIELD_A, t, island)
                if u is not None:
                    ut = np.asarray(u[t, island])
                else:
                    ut = orc.philox_resample_uniforms(pf.seed, scheme, N, t, island)
                if scheme == "multinomial":
                    # the tape's sorted uniforms, or -- production mode -- the device's own
                    # (smc_filter_spacings: the step loop never writes them; statistics and the Philox
                    # layout of the draws are checked in check_device_spacings)
                    su = ut if u is not None else pf._spacings(t, island)
                    ut = su
                else:
                    su = orc.sorted_uniforms(scheme, N, ut)
                W_dev = pf._history(_lib.FIELD_W, t - 1, island)
                if su is not None:
                    if two_level:
                        A_c, _ = orc.inverse_cdf_2level_c(scheme, ut, lw_prev)
                    else:
                        A_c = orc.inverse_cdf_q62(su, W_dev)
                    assert np.array_equal(A, A_c), (t, int(np.sum(A != A_c)))      # the contract: bit-exact
                    W_ref = orc.exp_and_normalise(lw_prev)
                    try:
                        A_ref = orc.inverse_cdf(su, W_ref)
                    except IndexError:
                        A_ref = None
                    if A_ref is not None and not np.array_equal(A, A_ref):
                        n, ok = orc.audit_near_ties(su, W_ref, A_ref, A)
                        assert ok, (t, n)
                        near_ties += n
                assert A.min() >= 0 and A.max() < N and np.all(np.diff(A) >= 0)
                Xp = X_prev[A]
            else:
                Xp = X_prev
        # ---- move and weigh (core.py:315-324) from the device's own parents
        d = X.shape[1] if X.ndim == 2 else 1
        if z is not None:
            zt = np.asarray(z[t, island])
        elif d == 1:
            zt = orc.philox_normals(pf.seed, N, t, island)
        else:
            zt = orc.philox_normals_mv(pf.seed, N, d, t, island)
        if zt is not None:
            Xo, inc = orc.propagate(model, fk, t, np.asarray(y[t]), Xp, zt, ctx)
            lwo = inc if (t == 0 or rs_flag) else lw_prev + inc
            lwo = np.where(np.isnan(lwo), -np.inf, lwo)
            if exact and z is not None:
                assert np.array_equal(X, Xo), (t, "X")
                if fk == "bootstrap":
                    assert np.array_equal(lw, lwo), (t, "lw")
                else:
                    assert np.allclose(lw, lwo, rtol=tol, atol=tol), (t, "lw")
            else:
                assert np.allclose(X, Xo, rtol=tol, atol=tol), (t, "X", float(np.max(np.abs(X - Xo))))
                assert np.allclose(lw, lwo, rtol=100 * tol, atol=100 * tol), (t, "lw")
        # ---- summaries (resampling.py:217-226, core.py:351-359) from the device's log-weights
        w = orc.Weights(lw=lw.copy())
        assert abs(summ[t, 0] / w.ESS - 1) < 1e-10, (t, "ESS")
        assert abs(summ[t, 1] - w.log_mean) < 1e-11 * max(1.0, abs(w.log_mean)), (t, "log_mean")
        loglt = w.log_mean if (t == 0 or rs_flag) else w.log_mean - prev_log_mean
        assert abs(summ[t, 2] - loglt) < 1e-10 * max(1.0, abs(loglt)), (t, "loglt")
        if steps is None:
            logLt += summ[t, 2]
            assert abs(summ[t, 3] - logLt) < 1e-10 * max(1.0, abs(logLt)), (t, "logLt")
        prev_log_mean = w.log_mean
        lw_prev, X_prev = lw, X
    return near_ties


EXACT_MODELS = ("toy", "lg", "gordon")          # IEEE + - * / only: X (and bootstrap lw) bit-exact

NEAR_TIE_LOG = []


# The integer two-level CDF contract and the reference's sequential fp64 CDF choose a different -- certified near-tie --
# ancestor about 6e-8 times per ancestor (28 in 4.6e8 audited, GPUTEST r5): a run is held to 20 times that rate
# (round 5 allowed 1e-5: a regression of two orders of magnitude would have passed), and to at least one.
NEAR_TIE_RATE = 6e-8


def near_tie_allowance(draws):
    return max(1, int(20 * NEAR_TIE_RATE * draws))


def log_near_ties(where, ties, draws):
    """Every audit records how many ancestors differed from the reference's sequential fp64 CDF
    (each one certified a near-tie) out of how many draws: the test log then holds the observed
    rate (pytest -s / the captured output of the GPU log), not an expectation."""
    NEAR_TIE_LOG.append((where, int(ties), int(draws)))
    print("near-ties vs the reference CDF: %-46s %6d of %12d ancestors (%.2e)"
          % (where, ties, draws, ties / max(1, draws)))


def check_filter_replay(golden, case, model, fk, T=None, N=None):
    """Replay the reference's own draws through the fused device loop.  ``N`` overrides
    the fixture's population size (the oracle then plays the reference's part).

    Three layers: (1) the free-running reference-semantics oracle: same branch at every step,
    ESS / log-evidence to 1e-9; (2) ``audit_history``: every step teacher-forced from the
    device's own previous state -- ancestors bit-exact against the device's integer CDF
    contract, every mismatch against the reference's sequential CDF certified a near-tie, X /
    lw of EVERY particle compared (bit-exact for the IEEE-only models); (3) the production
    kernels (no history: slot parity baked into the launches) give the same bits as the
    history-keeping ones that were audited."""
    g = golden(case)
    mk_dev, mk_orc = MODELS[model]
    scheme, ESSrmin = str(g["scheme"]), float(g["ESSrmin"])
    N = int(g["N"]) if N is None else N
    y = list(g["y"])[:T] if T else list(g["y"])
    np.random.seed(int(g["run_seed"]))
    rec = orc.RecordingRNG()
    o = orc.run_filter(mk_orc(), y, N, scheme, ESSrmin, fk=fk, rng=rec, keep=True)
    if T is None and N == int(g["N"]):       # the oracle reproduces the reference bit for bit (pinned)
        assert o["final_logLt"] == float(g["logLt"])
    z, u = tapes_from_oracle(rec.tape, len(y), N, scheme)
    cls = ssm.Bootstrap if fk == "bootstrap" else ssm.GuidedPF
    mk = lambda hist: pa.SMC(fk=cls(ssm=mk_dev(), data=y), N=N, resampling=scheme, ESSrmin=ESSrmin,
                             replay=(z, u), store_history=hist)
    pf = mk(False)
    pf.run()
    assert pf.summaries.rs_flags == o["rs_flag"]                       # same branch every step
    assert rel(pf.summaries.ESSs, o["ESS"]) < 1e-9
    assert rel(pf.summaries.logLts, o["logLt"]) < 1e-9                 # north star: 1e-6
    assert abs(pf.logLt / o["final_logLt"] - 1) < 1e-9
    ph = mk(True)
    ph.run()
    assert np.array_equal(ph.X, pf.X) and np.array_equal(ph.wgts.lw, pf.wgts.lw)
    assert np.array_equal(ph._summ(), pf._summ())
    if len(y) > 1:
        assert np.array_equal(ph.A, pf.A)
    mv = model.startswith("mv")
    ties = audit_history(ph, mk_orc, fk, y, scheme, ESSrmin, z=z, u=u,
                         exact=model in EXACT_MODELS, tol=1e-11 if mv else 1e-12)
    log_near_ties("replay %s/%s N=%d T=%d %s" % (case, fk, N, len(y), scheme), ties, sum(o["rs_flag"]) * N)
    assert ties <= near_tie_allowance(len(y) * N), ties
    if ties == 0 and not mv and len(y) > 1 and np.array_equal(pf.A, o["A"]):
        # nothing flipped anywhere: the free-running oracle IS this run
        exact = model.startswith(EXACT_MODELS)
        assert np.max(np.abs(pf.X - o["X"])) <= (0 if exact else 1e-12)
        assert rel(pf.W, o["W"]) < 1e-10
    return pf, o


def check_model_philox_vs_oracle(golden, case, model, N=20000, runs=16):
    """Production (Philox) mode of a nonlinear model: no exact likelihood exists, so the
    device's log-evidence estimates are compared with the oracle's (numpy RNG) on the
    same data -- both are unbiased-in-L estimators of the same quantity: `runs` independent
    runs on each side and a two-sample z-test on the means (|z| < 4: a false alarm once in
    16 000 test runs; a shift of one Monte-Carlo sd of a single run is a 2.8-sigma event at
    16 runs), plus an F-type bound on the spreads (the device's estimator must not be
    noisier than the reference's: variance ratio within [1/6, 6], the 99.9 % band of F(15, 15))."""
    g = golden(case)
    mk_dev, mk_orc = MODELS[model]
    y = list(g["y"])
    dev, ref = [], []
    for s in range(runs):
        pf = pa.SMC(fk=ssm.Bootstrap(ssm=mk_dev(), data=y), N=N, seed=40 + s)
        pf.run()
        dev.append(pf.logLt)
        np.random.seed(900 + s)
        ref.append(orc.run_filter(mk_orc(), y, N, "systematic", 0.5)["final_logLt"])
    dev, ref = np.array(dev), np.array(ref)
    assert np.all(np.isfinite(dev)) and len(set(dev.tolist())) == runs
    se = np.sqrt(dev.var(ddof=1) / runs + ref.var(ddof=1) / runs)
    z = (dev.mean() - ref.mean()) / se
    assert abs(z) < 4.0, (case, z, dev.mean(), ref.mean(), se)
    ratio = dev.var(ddof=1) / ref.var(ddof=1)
    lim = 6.0 if runs >= 16 else 12.0
    assert 1.0 / lim < ratio < lim, (case, ratio)


def check_two_level_cdf(golden, monkeypatch, N=4096, T=20):
    """N = 2^k with 2..1024 tiles, systematic / stratified: the step loop runs on the two-level
    exact CDF (k_ancestors2).  Replay of the reference's draws: the oracle run on the same
    contract (cdf="2level": orc_inverse_cdf_2level + the contract's ESS) must be THE SAME RUN,
    bit for bit -- ancestors, particles, log-weights, decisions; so must the device variants
    (fp64 band shortcut off; k_reduce2 in front); the reference-semantics oracle and the flat-Q62
    device path agree up to audited near-ties (check_filter_replay does the audit per step)."""
    for case, scheme in (("toy_systematic", "systematic"), ("toy_stratified", "stratified"),
                         ("toy_multinomial", "multinomial")):
        g = golden(case)
        mk_dev, mk_orc = MODELS["toy"]
        y = list(g["y"])[:T]
        np.random.seed(int(g["run_seed"]))
        rec = orc.RecordingRNG()
        o = orc.run_filter(mk_orc(), y, N, scheme, 0.5, rng=rec, keep=True)          # reference semantics
        z, u = tapes_from_oracle(rec.tape, len(y), N, scheme)
        o2 = orc.run_filter(mk_orc(), y, N, scheme, 0.5, rng=orc.ReplayRNG(rec.tape), cdf="2level",
                            keep=True)
        runs = {}
        for name, env in (("two_level", {}), ("exact_counts", {"SMC_EXACT_COUNTS": "1"}),
                          ("mid", {"SMC_TWO_LEVEL_MID": "1"}), ("flat", {"SMC_FLAT_CDF": "1"}),
                          ("narrow", {"SMC_NO_WIDE": "1"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            pf = pa.SMC(fk=ssm.Bootstrap(ssm=mk_dev(), data=y), N=N, resampling=scheme, ESSrmin=0.5,
                        replay=(z, u), store_history=(name == "two_level"))
            if name in ("two_level", "narrow"):         # (multinomial: k_reduce2 in front, one tile per workgroup)
                assert ("k_ancestors2w" in describe(pf)) == (name != "narrow" and scheme != "multinomial"), describe(pf)
            pf.run()
            runs[name] = (np.array(pf.A), np.array(pf.X), list(pf.summaries.logLts), list(pf.summaries.rs_flags))
            if name == "two_level":
                assert "k_ancestors2" in describe(pf)
                # the oracle on the device's contract is the device's run, step by step
                for t in range(len(y)):
                    assert np.array_equal(pf.hist.X[t], o2["hist"]["X"][t]), t
                    assert np.array_equal(pf.hist.wgts[t].lw, o2["hist"]["lw"][t]), t
                    if o2["rs_flag"][t]:
                        assert np.array_equal(pf.hist.A[t], o2["hist"]["A"][t]), t
                assert list(pf.summaries.rs_flags) == o2["rs_flag"]
                assert rel(pf.summaries.logLts, o2["logLt"]) < 1e-12
            monkeypatch.undo()
        A2, X2, ll2, rf2 = runs["two_level"]
        assert np.array_equal(A2, runs["exact_counts"][0]) and np.array_equal(X2, runs["exact_counts"][1])
        assert ll2 == runs["exact_counts"][2]
        # k_reduce2 in front (one workgroup per island reduces the partials) instead of every
        # workgroup of k_ancestors2: the same operations in the same order, the same bits
        assert np.array_equal(A2, runs["mid"][0]) and np.array_equal(X2, runs["mid"][1])
        assert ll2 == runs["mid"][2]
        # one tile per workgroup (k_ancestors2) / two (the default, k_ancestors2w): the same bits
        for name in ("narrow",):
            assert np.array_equal(A2, runs[name][0]) and np.array_equal(X2, runs[name][1]) and ll2 == runs[name][2], name
        assert rf2 == o["rs_flag"] and any(rf2)
        assert rel(ll2, o["logLt"]) < 1e-9 and rel(runs["flat"][2], o["logLt"]) < 1e-9
    check_two_level_injected()


def check_wide_general(monkeypatch, sizes=(1500, 3000, 17 * 1024 + 1, 19 * 1024 - 3), T=6):
    """Any N under the systematic scheme on k_ancestors2w<.., POW2 = false> (two tiles per workgroup, the general
    counts, odd numbers of tiles, a ragged last tile, runs of tiles per XCD) against k_ancestors2 (one tile per
    workgroup): the same run bit for bit -- typical weights and a collapsing population (heavy parents), one island
    and three, with the fp64 shortcut off and with the plain tile map."""
    for N in sizes:
        for sigY in (0.2, 1e-4):
            rng = np.random.RandomState(N % 1000)
            x = np.cumsum(rng.standard_normal(T))
            y = [np.array([v]) for v in x + sigY * rng.standard_normal(T)]
            model = kalman.LinearGauss(sigmaY=sigY, sigmaX=1.0, rho=0.9, sigma0=1.0)
            runs = {}
            for name, env in (("narrow", {"SMC_NO_WIDE": "1"}), ("wide", {}), ("exact", {"SMC_EXACT_COUNTS": "1"}),
                              ("plain_map", {"SMC_NO_XCD_CHUNKS": "1"})):
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                pf = pa.SMC(fk=ssm.Bootstrap(ssm=model, data=y), N=N, resampling="systematic", ESSrmin=1.0,
                            collect="off", seed=5, store_history=True, n_islands=3 if N < 4000 else 1)
                assert ("k_ancestors2w" in describe(pf)) == (name != "narrow"), (name, describe(pf))
                pf.run()
                runs[name] = ([np.array(a) for a in pf.hist.A[1:]], np.array(pf.X), np.array(pf.logLt))
                monkeypatch.undo()
            for name in ("wide", "exact", "plain_map"):
                assert all(np.array_equal(a, b) for a, b in zip(runs["narrow"][0], runs[name][0])), (N, sigY, name)
                assert np.array_equal(runs["narrow"][1], runs[name][1]) and np.array_equal(runs["narrow"][2], runs[name][2]), (N, sigY, name)


def check_two_level_injected(sizes=(4096, 3000)):
    """The contract itself on weights no filter run would produce -- skewed, -inf entries, an
    empty tile, a collapsed vector: uploaded with smc_filter_set_state, one resampling step on the
    device, ancestors against orc_inverse_cdf_2level BIT FOR BIT and against the reference's
    sequential fp64 CDF (resampling.py:500-509) with every mismatch certified a near-tie."""
    y = [np.array([0.3]), np.array([0.1])]
    for N in sizes:
        rng = np.random.default_rng(12 + N)
        cases = []
        lw = rng.normal(0.0, 3.0, size=N)
        cases.append(("skewed", lw.copy()))
        lw[rng.random(N) < 0.05] = -np.inf
        lw[1024:2048] = -np.inf                                  # an empty tile
        cases.append(("minus_inf", lw.copy()))
        cases.append(("very_skewed", rng.normal(0.0, 40.0, size=N)))
        c = np.full(N, -800.0)
        c[N // 3] = 0.0
        cases.append(("collapsed", c))
        cases.append(("flat", np.zeros(N)))
        # exactly summable weights on the two-level (headline) path -- SURVEY hard part 1's known-answer test:
        # 2^m particles of weight exp(0) = 1 (p = 1, k = 0 exactly in the (p, k) form) at random positions, the
        # others -inf: W = 2^-m exactly, every partial sum of the reference's loop and every quantity of the
        # contract (S_b, the shares n_b 2^(52-m), c_j Q_b / t_b = j 2^(52-m)) is exact, so the ancestors must be
        # the reference's (resampling.py:500-509) 100 % -- no near-tie allowance
        m_alive = 1 << (int(np.log2(N // 2)))
        lwd = np.full(N, -np.inf)
        lwd[rng.choice(N, size=m_alive, replace=False)] = 0.0
        cases.append(("dyadic", lwd))
        for scheme in ("systematic", "stratified"):
            for name, lwi in cases:
                pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=N, resampling=scheme,
                            ESSrmin=2.0, seed=7, collect="off")         # ESSrmin 2: always resample
                next(pf)
                pf.set_state(lw=lwi)
                assert np.array_equal(pf.wgts.lw, lwi)
                red = orc.two_level_reduce(*orc.tile_partials(lwi))[0]
                assert pf.wgts.ESS == red["ESS"], (name, pf.wgts.ESS, red["ESS"])
                X0 = np.array(pf.X)
                next(pf)
                assert pf.rs_flag
                ut = orc.philox_resample_uniforms(pf.seed, scheme, N, 1)
                A_c, _ = orc.inverse_cdf_2level_c(scheme, ut, lwi)
                A = np.array(pf.A)
                assert np.array_equal(A, A_c), (N, scheme, name, int(np.sum(A != A_c)))
                assert np.array_equal(pf.Xp, X0[A])
                su = orc.sorted_uniforms(scheme, N, ut)
                W = orc.exp_and_normalise(lwi)
                assert np.all(W[A] > 0)
                A_ref = orc.inverse_cdf(su, W)
                n, ok = orc.audit_near_ties(su, W, A_ref, A)
                log_near_ties("injected %s %s N=%d" % (name, scheme, N), n, N)
                assert ok and n <= max(1, N // 100000), (N, scheme, name, n)
                if name == "dyadic":
                    assert np.array_equal(A, A_ref), (N, scheme, int(np.sum(A != A_ref)))
        # multinomial: the sorted uniforms come from a tape (counts are searches over them), or --
        # production mode -- are the device's own draws, regenerated tile by tile inside k_ancestors2
        # (smc_filter_spacings writes them out for the oracle; with the fp64 shortcut and without)
        z = rng.standard_normal((2, 1, N))
        su_tape = orc.uniform_spacings_from(rng.random(N + 1))
        u = np.stack([np.zeros(N), su_tape]).reshape(2, 1, N)
        for (name, lwi), rep in [((n_, l_), r_) for n_, l_ in cases for r_ in (True, False)]:
            pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=N, resampling="multinomial",
                        ESSrmin=2.0, seed=7, collect="off", replay=(z, u) if rep else None)
            assert "k_ancestors2" in describe(pf) and ("k_f_spacing_onepass" in describe(pf)) == (not rep)
            next(pf)
            pf.set_state(lw=lwi)
            X0 = np.array(pf.X)
            next(pf)
            assert pf.rs_flag
            su = su_tape if rep else pf._spacings(1)
            A_c, _ = orc.inverse_cdf_2level_c("multinomial", su, lwi)
            A = np.array(pf.A)
            assert np.array_equal(A, A_c), (N, "multinomial", name, int(np.sum(A != A_c)))
            assert np.array_equal(pf.Xp, X0[A])
            W = orc.exp_and_normalise(lwi)
            assert np.all(W[A] > 0)
            n, ok = orc.audit_near_ties(su, W, orc.inverse_cdf(su, W), A)
            log_near_ties("injected %s multinomial %s N=%d" % (name, "tape" if rep else "philox", N), n, N)
            assert ok and n <= max(1, N // 100000), (N, "multinomial", name, n)
            if name == "dyadic":
                assert np.array_equal(A, orc.inverse_cdf(su, W)), (N, "multinomial", rep)


def check_oracle_at_size(model, mk_dev, mk_orc, N, T, scheme="systematic", ESSrmin=0.5, fk="bootstrap",
                         replay=True, n_islands=1, islands=(0,), seed=5, d=1, data_seed=3,
                         expect_resample=True, dy=None):
    """Oracle parity at BASELINE.json's sizes: a store_history run of T steps audited step by
    step (audit_history: ancestors bit-exact against the contract, near-ties against the
    reference's CDF certified, every particle's X / lw, ESS, evidence), and the production
    kernels (no history) reproduce its final state bit for bit.  replay=True: the numpy draws
    of a reference run of this shape are generated here and fed through the tapes (X and, for
    the IEEE-only models, lw then are bit-exact); False: production Philox streams."""
    rng = np.random.RandomState(data_seed)
    if d == 1:
        y = [np.array([v]) for v in 0.4 * np.cumsum(rng.standard_normal(T))]
    else:
        y = [rng.standard_normal((1, dy or d)) for _ in range(T)]
    cls = ssm.Bootstrap if fk == "bootstrap" else ssm.GuidedPF
    z = u = None
    if replay:
        assert n_islands == 1
        # numpy.random's legacy generator, drawn directly into the dense tapes (any tapes are
        # valid inputs: the claim is "identical normals and uniforms in, identical particles out")
        np.random.seed(seed)
        z = np.random.standard_normal((T, 1, N) if d == 1 else (T, 1, N, d))
        K = 1 if scheme == "systematic" else N
        u = np.random.rand(T, 1, K)
        if scheme == "multinomial":
            for t in range(T):
                u[t, 0] = orc.uniform_spacings_from(np.random.rand(N + 1))
    mk = lambda hist: pa.SMC(fk=cls(ssm=mk_dev(), data=y), N=N, resampling=scheme, ESSrmin=ESSrmin,
                             replay=None if z is None else (z, u), store_history=hist,
                             n_islands=n_islands, seed=seed, collect="off")
    ph = mk(True)
    ph.run()
    ties = 0
    for isl in islands:
        ties += audit_history(ph, mk_orc, fk, y, scheme, ESSrmin, z=z, u=u, exact=model.startswith(EXACT_MODELS),
                              island=isl, tol=1e-11 if d > 1 else 1e-12)
    log_near_ties("%s/%s N=%d T=%d %s %s" % (model, fk, N, T, scheme, "replay" if replay else "philox"), ties,
                  int(np.sum(ph._summ()[list(islands), :, 4])) * N)
    assert ties <= near_tie_allowance(len(islands) * T * N), ties
    pf = mk(False)
    pf.run()
    for isl in islands:
        assert np.array_equal(pf._get(_lib.FIELD_X, isl), ph._get(_lib.FIELD_X, isl))
        assert np.array_equal(pf._get(_lib.FIELD_LW, isl), ph._get(_lib.FIELD_LW, isl))
        if ph._summ()[isl, -1, 4]:
            assert np.array_equal(pf._get(_lib.FIELD_A, isl), ph._get(_lib.FIELD_A, isl))
    assert np.array_equal(pf._summ(), ph._summ())
    assert not expect_resample or np.any(ph._summ()[:, 1:, 4] != 0)
    return ties


def check_device_spacings(sizes=(2048, 3000, 1 << 14), seed=91):
    """The sorted uniforms of the multinomial resampling in production mode (uniform_spacings,
    resampling.py:512-537, by exponential spacings drawn from the Philox stream): smc_filter_spacings
    against the oracle's restatement of the draw layout (draw n <- word n & 1 of call n >> 1, stream 2,
    q_n = rint(-log(u_n) 2^s), su_n = Z_n / Z_N) -- equal up to the ulps by which the device's
    table-driven log differs from libm's inside the rint -- sorted, in (0, 1), uniform order statistics;
    and the ancestors of the step are the contract's for exactly these uniforms (audit_history)."""
    rng = np.random.RandomState(4)
    T = 6
    y = [np.array([v]) for v in 0.5 * np.cumsum(rng.standard_normal(T))]
    for N in sizes:
        for n_islands, isl in ((1, 0), (3, 2)):
            pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=N, resampling="multinomial", ESSrmin=1.0,
                        seed=seed, store_history=True, n_islands=n_islands, collect="off")
            pf.run()
            for t in (1, T - 1):
                su = pf._spacings(t, isl)
                want = orc.philox_spacings(seed, N, t, isl)
                assert su.shape == (N,) and np.all(np.diff(su) >= 0) and su[0] > 0 and su[-1] < 1
                assert np.max(np.abs(su - want)) < 1e-13, (N, t, float(np.max(np.abs(su - want))))
            ties = audit_history(pf, lambda: orc.ToySSM(0.2), "bootstrap", y, "multinomial", 1.0, island=isl)
            assert ties <= near_tie_allowance(T * N), ties
    # order statistics of N uniforms: E su_n = (n + 1) / (N + 1), and a Kolmogorov distance of the size
    # 1 / sqrt(N) from the uniform law
    N = sizes[-1]
    d = np.max(np.abs(su - (np.arange(N) + 1.0) / (N + 1)))
    assert d < 2.5 / np.sqrt(N), d


def check_strict_ancestors(sizes=(3000, 4096), op_N=1 << 14, op_cases=12, schemes=("systematic", "stratified", "multinomial"),
                           model="toy", small=True, replays=(True, False), T=7, ESSrmin=0.9):
    """The literal guarantee of the north star: identical (su, W) in, the reference's ancestors out.
    (1) operator: rs.inverse_cdf(su, W, strict=True) == the reference's sequential loop
    (resampling.py:500-509, restated in the oracle and pinned by the golden fixtures) on random,
    skewed, sparse and near-degenerate weight vectors, no tolerance, no near-tie audit;
    (2) the fused loop with strict_ancestors=True: at every resampling step
    A_t == inverse_cdf(su_t, W_{t-1}) with W_{t-1} the filter's own weights, for the three schemes,
    replayed and Philox draws, N <= 1024 (flat step) and above (two-level step), islands."""
    rng = np.random.default_rng(5)
    for c in range(op_cases):
        N = op_N if c % 3 else op_N + 37
        kind = c % 4
        lw = rng.normal(0.0, (0.5, 3.0, 12.0, 40.0)[kind], size=N)
        if kind == 2:
            lw[rng.random(N) < 0.3] = -np.inf
        W = orc.exp_and_normalise(lw)
        for su in ((rng.random() + np.arange(N)) / N, (rng.random(N) + np.arange(N)) / N,
                   orc.uniform_spacings_from(rng.random(N + 1))):
            try:
                want = orc.inverse_cdf(su, W)
            except IndexError:
                continue                                   # su[-1] beyond the rounded total: no reference answer
            got = rs.inverse_cdf(su, W, strict=True)
            assert np.array_equal(got, want), (c, int(np.sum(got != want)))
    # rs.set_strict: the schemes themselves
    if op_cases:
        rs.set_strict(True)
        try:
            W = orc.exp_and_normalise(rng.normal(0.0, 2.0, size=5000))
            for scheme in ("systematic", "stratified", "multinomial"):
                np.random.seed(8)
                got = rs.resampling(scheme, W, M=4000)
                np.random.seed(8)
                want = orc.resampling(scheme, W, M=4000)
                assert np.array_equal(got, want), scheme
        finally:
            rs.set_strict(False)
    # (2)
    yr = np.random.RandomState(2)
    y = [np.array([v]) for v in 0.4 * np.cumsum(yr.standard_normal(T))]
    # ("peaky" / "collapsed": a few parents -- one parent -- own most offspring: the scatter search's multi-window passes)
    mk_model = {"toy": lambda: kalman.ToySSM(0.2), "sv": lambda: ssm.StochVol(), "peaky": lambda: kalman.ToySSM(2e-3),
                "collapsed": lambda: kalman.ToySSM(1e-6)}[model]
    for N in ((700,) if small else ()) + tuple(sizes):
        for scheme in schemes:
            for replay in replays:
                z = u = None
                if replay:
                    np.random.seed(3)
                    z = np.random.standard_normal((T, 1, N))
                    u = np.random.rand(T, 1, 1 if scheme == "systematic" else N)
                    if scheme == "multinomial":
                        for t in range(T):
                            u[t, 0] = orc.uniform_spacings_from(np.random.rand(N + 1))
                nisl = 1 if replay else 2
                pf = pa.SMC(fk=ssm.Bootstrap(ssm=mk_model(), data=y), N=N, resampling=scheme, ESSrmin=ESSrmin,
                            seed=11, store_history=True, strict_ancestors=True, collect="off",
                            replay=None if z is None else (z, u), n_islands=nisl)
                pf.run()
                d = describe(pf)                            # (the sequential CDF by its parallel emulation, csrc/smc_seqx.h)
                assert ("k_strict_classify+k_strict_search" in d) if N > 1024 else ("k_sqx_classify" in d), d
                summ = pf._summ()
                for isl in range(nisl):
                    nres = 0
                    for t in range(1, T):
                        if not summ[isl, t, 4]:
                            continue
                        nres += 1
                        W = pf._history(_lib.FIELD_W, t - 1, isl)
                        if u is not None:
                            ut = np.asarray(u[t, isl])
                        elif scheme == "multinomial":
                            ut = pf._spacings(t, isl)
                        else:
                            ut = orc.philox_resample_uniforms(pf.seed, scheme, N, t, isl)
                        su = ut if scheme == "multinomial" else orc.sorted_uniforms(scheme, N, ut)
                        try:
                            want = orc.inverse_cdf(su, W)
                        except IndexError:
                            continue
                        A = pf._history(_lib.FIELD_A, t, isl)
                        assert np.array_equal(A, want), (N, scheme, replay, t, int(np.sum(A != want)))
                        assert np.array_equal(pf._history(_lib.FIELD_XP, t, isl), pf._history(_lib.FIELD_X, t - 1, isl)[A])
                    assert nres >= 2, (N, scheme, nres)
                    ex, nx = ctypes.c_int64(-1), ctypes.c_int64(-1)       # (the last step took the two-launch path, verified)
                    _lib.check(_lib.lib().smc_filter_strict_stats(pf._f, isl, ctypes.byref(ex), ctypes.byref(nx)))
                    assert ex.value == 0 and 0 < nx.value <= 200, (N, scheme, ex.value, nx.value)
                    log_near_ties("STRICT %s %s N=%d %s isl %d: literal equality" % (model, scheme, N, "replay" if replay else "philox", isl),
                                  0, nres * N)
    if not small:
        return
    with pytest.raises(ValueError):
        pa.SMC(fk=ssm.AuxiliaryPF(ssm=ssm.StochVol(), data=y), N=500, strict_ancestors=True)


def check_strict_never_leaves_the_fast_path(cases, T=300):
    """The two-launch emulation must not merely be right (its exact path makes it right whatever happens): on a filter's
    own weights it must verify EVERY step.  Round 5's soak found one step in five hundred taking the exact path
    (milliseconds at N = 2^20): the estimate of the sum in front of a thread was formed as `inclusive - own`, which
    cancels at the head of every array, where the weights rise over a hundred binades in a dozen elements.  Every
    step's statistics are read here, not the last one's."""
    yr = np.random.RandomState(4)
    y = [np.array([v]) for v in 0.4 * np.cumsum(yr.standard_normal(T))]
    mk = {"toy": lambda: kalman.ToySSM(0.2), "sv": lambda: ssm.StochVol(), "peaky": lambda: kalman.ToySSM(2e-3),
          "collapsed": lambda: kalman.ToySSM(1e-6)}
    for N, nisl, scheme, model, ESSrmin in cases:
        pf = pa.SMC(fk=ssm.Bootstrap(ssm=mk[model](), data=y), N=N, resampling=scheme, ESSrmin=ESSrmin, seed=77,
                    strict_ancestors=True, collect="off", n_islands=nisl)
        worst = 0
        for t in range(T):
            pf.step_async(1)
            for isl in range(nisl):
                ex, nx = ctypes.c_int64(-1), ctypes.c_int64(-1)
                _lib.check(_lib.lib().smc_filter_strict_stats(pf._f, isl, ctypes.byref(ex), ctypes.byref(nx)))
                assert ex.value == 0, (N, nisl, scheme, model, t, isl, ex.value, nx.value)
                worst = max(worst, nx.value)
        assert np.all(np.isfinite(pf.logLts_islands)) and worst <= 256, (N, scheme, model, worst)


def check_strict_operator_path(monkeypatch, sizes=(1500, 1 << 13), T=5):
    """strict_ancestors=True OUTSIDE the fused strict step (strict with qmc): the template-method
    step of a user-level FeynmanKac and SQMC -- whose inverse_cdf sees the weights of particles in SORTED order, a
    likelihood over ordered states: the running sum climbs through hundreds of binades, the shape that sends
    smc_inverse_cdf_strict beyond its exception lists to its exact path.  Every call of inverse_cdf the run makes is
    recorded: it must have been made in strict mode and return resampling.py:484-509's ancestors literally."""
    calls = []
    real = rs.inverse_cdf

    def spy(su, W, strict=None):
        A = real(su, W, strict=strict)
        calls.append((np.asarray(su).copy(), np.asarray(W).copy(), np.asarray(A).copy(),
                      bool(rs.STRICT[0] if strict is None else strict)))
        return A

    monkeypatch.setattr(rs, "inverse_cdf", spy)
    rng = np.random.RandomState(3)
    y = [np.array([v]) for v in 0.5 * np.cumsum(rng.standard_normal(T))]
    for N in sizes:
        for kw in (dict(qmc=True), dict(qmc=True, sigY=2e-3), dict(resampling="stratified", ESSrmin=1.0),
                   dict(resampling="multinomial", ESSrmin=1.0)):
            kw = dict(kw)
            model = kalman.ToySSM(kw.pop("sigY", 0.2))
            fk = ssm.Bootstrap(ssm=model, data=y) if kw.get("qmc") else _PickleCustomFK(ssm=model, data=y)
            del calls[:]
            np.random.seed(5)
            n = 1 << int(np.log2(N)) if kw.get("qmc") else N          # (Sobol' points: N = 2^k)
            pf = pa.SMC(fk=fk, N=n, strict_ancestors=True, collect="off", **kw)
            assert not pf._fused                       # (numpy draws / a user subclass: device operators)
            pf.run()
            assert len(calls) == T - 1 and not rs.STRICT[0], (N, kw, len(calls))
            for su, W, A, strict in calls:
                assert strict and np.array_equal(A, orc.inverse_cdf(su, W)), (N, kw)
            assert np.isfinite(pf.logLt)


def check_seq_prefix_sums(sizes=(5000, 1 << 14, 20001), monkeypatch=None):
    """csrc/smc_seqsum.h: the reference's sequential fp64 prefix sums (resampling.py:506-508: s = W[0]; s += W[j])
    computed in parallel must be THE SAME DOUBLES as the loop's, whatever the weights: the element-level pass (mode 0),
    the tile walk it falls back to (mode 2) and the literal one-lane walk (mode 1) against a Python loop -- random,
    uniform, skewed over 100 orders of magnitude, one particle holding all the mass (zeros around a power of two),
    sparse, exactly summable, engineered rounding ties, subnormal-range and unnormalised weights.  And the filter's
    strict mode gives the same run with the literal walk in place of the emulation."""
    def seq(W, mode):
        d = DeviceArray.from_numpy(np.ascontiguousarray(W))
        S = DeviceArray((len(W),))
        c = ctypes.c_int64(-9)
        _lib.check(_lib.lib().smc_seq_prefix_sums(_lib.ctx().h, d.ptr, len(W), S.ptr, mode, ctypes.byref(c)))
        return S.get(), c.value

    def loop(W):
        out = np.empty_like(W)
        acc = W[0]
        out[0] = acc
        for j in range(1, len(W)):
            acc = acc + W[j]
            out[j] = acc
        return out

    rng = np.random.default_rng(1)
    fast = 0
    log = []
    for N in sizes:
        cases = {}
        w = np.exp(3 * rng.standard_normal(N)); cases["lognormal"] = w / w.sum()
        cases["uniform"] = np.full(N, 1.0 / N)
        w = np.exp(40 * rng.standard_normal(N)); cases["very skewed"] = w / w.sum()
        w = np.zeros(N); w[N // 3] = 1.0; cases["collapsed"] = w
        w = rng.random(N); w[rng.random(N) < 0.3] = 0.0; cases["sparse"] = w / w.sum()
        cases["dyadic"] = rng.integers(0, 2 ** 30 // N, size=N).astype(np.float64) / 2.0 ** 30
        w = np.full(N, 2.0 ** -40); w[0] = 0.25; w[1::2] = 3 * 2.0 ** -55; cases["ties"] = w
        w = np.full(N, 2.0 ** -40); w[0] = 0.25; w[1:241:2] = 3 * 2.0 ** -55; cases["some ties"] = w    # 120 ties: the walk's LDS form
        cases["tiny"] = rng.random(N) * 1e-300
        cases["unnormalised"] = rng.random(N) * 1e6
        w = np.zeros(N); w[-1] = 0.5; cases["late mass"] = w
        # what a filter resamples on: a few heavy particles, the rest negligible -- the running sum sits just
        # below 1.0 for most of the array
        w = np.exp(-60.0 + rng.standard_normal(N)); w[rng.choice(N, 8, replace=False)] = 1.0; cases["degenerate"] = w / w.sum()
        w = np.full(N, 2.0 ** -44); w[:4] = 0.25 - N * 2.0 ** -46; cases["trailing"] = w
        # a likelihood over SORTED states (what SQMC's Hilbert order hands to inverse_cdf, core.py:339-349): the running sum
        # rises through several hundred binades, every element of the flank an exception -- more than the lists hold: the
        # exact path (same doubles, milliseconds)
        x = np.sort(rng.standard_normal(N)) * 1.3
        w = np.exp(-0.5 * ((x - 0.3) / 0.2) ** 2); cases["sorted gaussian"] = w / w.sum()
        for name, W in cases.items():
            want = loop(W).view(np.uint64)
            a, fb = seq(W, 0)
            assert np.array_equal(a.view(np.uint64), want), (N, name, "element pass", int((a.view(np.uint64) != want).sum()))
            b, nx = seq(W, 2)
            assert np.array_equal(b.view(np.uint64), want), (N, name, "tile walk", int((b.view(np.uint64) != want).sum()))
            c, _ = seq(W, 1)
            assert np.array_equal(c.view(np.uint64), want), (N, name, "literal")
            assert fb >= -1 and 0 < nx <= (N + 1023) // 1024
            fast += fb >= 0
            log.append((N, name, fb))
            if name not in ("ties", "tiny", "some ties", "sorted gaussian"):
                assert 0 <= fb <= 300, (N, name, fb)             # these stay on the fast path (fb: exceptions walked)
            if name == "some ties":
                assert 64 < fb <= 256, (N, name, fb)             # more exceptions than a wave holds, fewer than the list
            if name == "ties" and N >= 16384:
                assert fb == -1, (N, name)                       # more exceptions than the lists hold: the exact path
    assert fast >= 8 * len(sizes)
    if os.environ.get("SMC_TEST_VERBOSE"):
        print("seq prefix sums, exceptions walked (-1: exact path):", log)
    if monkeypatch is not None:
        y = [np.array([v]) for v in 0.4 * np.cumsum(np.random.RandomState(2).standard_normal(6))]
        runs = {}
        for name, env in (("emulated", {}), ("literal", {"SMC_STRICT_LITERAL": "1"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=3000, ESSrmin=0.9, seed=11, strict_ancestors=True,
                        collect="off", n_islands=2)
            pf.run()
            assert ("k_strict_cdf" in describe(pf)) == (name == "literal")
            runs[name] = (np.array(pf.A), np.array(pf.X), pf.logLts_islands.copy())
            monkeypatch.undo()
        assert all(np.array_equal(p, q) for p, q in zip(runs["emulated"], runs["literal"]))


def check_heavy_parents(monkeypatch, N=8192, T=12):
    """Collapsed / peaky weights: parents with >= 2048 offspring are registered by the ancestors
    kernel, which skips the blocks of offspring that are wholly theirs; k_propagate fills those.
    Same particles, ancestors and evidence as with the feature off, on the two-level and the flat
    path, N a power of two or not."""
    rng = np.random.RandomState(11)
    y = [np.array([v]) for v in np.cumsum(rng.standard_normal(T))]
    for sig, n in ((1e-6, N), (2e-3, N), (1e-6, N + 904)):
        mk = lambda: pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(sig), data=y), N=n, seed=4, ESSrmin=1.0)
        for flat in (False, True):
            runs = []
            for off in (False, True):
                if flat:
                    monkeypatch.setenv("SMC_FLAT_CDF", "1")
                if off:
                    monkeypatch.setenv("SMC_NO_HEAVY", "1")
                pf = mk()
                hist = []
                for _ in range(T):
                    next(pf)
                    hist.append(np.array(pf.A) if pf.rs_flag else None)
                runs.append((hist, np.array(pf.X), list(pf.summaries.logLts), list(pf.summaries.ESSs)))
                monkeypatch.undo()
            for a_on, a_off in zip(runs[0][0], runs[1][0]):
                assert (a_on is None) == (a_off is None)
                assert a_on is None or np.array_equal(a_on, a_off)
            assert np.array_equal(runs[0][1], runs[1][1]) and runs[0][2] == runs[1][2]
            if sig == 1e-6:                      # really collapsed: some step has one dominant parent
                cnt = max(np.bincount(a).max() for a in runs[0][0] if a is not None)
                assert cnt >= n // 2
    # several islands, each with its own list
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("SMC_NO_HEAVY", "1")
        pf = pa.SMC(fk=[ssm.Bootstrap(ssm=kalman.ToySSM(s_), data=y) for s_ in (1e-6, 0.2, 1e-5)], N=N, seed=9,
                    ESSrmin=1.0, collect="off")
        pf.run()
        outs.append((pf.logLts_islands.copy(), [pf._get(_lib.FIELD_X, k).copy() for k in range(3)],
                     [pf._get(_lib.FIELD_A, k).copy() for k in range(3)]))
        monkeypatch.undo()
    assert np.array_equal(outs[0][0], outs[1][0])
    for k in range(3):
        assert np.array_equal(outs[0][1][k], outs[1][1][k]) and np.array_equal(outs[0][2][k], outs[1][2][k])


def check_describe():
    """smc_filter_describe: which kernels a filter launches (the path selection DESIGN.md states)."""
    y = [np.array([0.1 * k]) for k in range(4)]

    def kernels(N, scheme="systematic", n_islands=1, model=None, **kw):
        fk = ssm.Bootstrap(ssm=model or kalman.ToySSM(0.2), data=y)
        pf = pa.SMC(fk=fk, N=N, resampling=scheme, n_islands=n_islands, seed=1, **kw)
        buf = ctypes.create_string_buffer(256)
        _lib.check(_lib.lib().smc_filter_describe(pf._f, buf, 256))
        return buf.value.decode()

    assert kernels(1000) == "k_filter_small"
    assert kernels(1 << 12) == "k_ancestors2w+k_propagate"                     # two-level, resident, N = 2^k: 2 tiles per workgroup
    assert kernels(1 << 12, "stratified") == "k_ancestors2w+k_propagate"
    assert kernels(1 << 12, n_islands=600) == "k_reduce2+k_ancestors2+k_propagate"   # 2400 workgroups
    assert kernels(3000) == "k_ancestors2w+k_propagate"                        # any N of >= 2 tiles, systematic: general counts,
    assert kernels(1500) == "k_ancestors2w+k_propagate"                        #  two tiles per workgroup (2 tiles, the second ragged)
    assert kernels(3000, "stratified") == "k_ancestors2+k_propagate"           # (the other closed-form scheme: one tile per workgroup)
    assert kernels(1 << 12, "multinomial") == \
        "k_f_spacing_onepass<with k_reduce2>+k_ancestors2+k_propagate"   # two-level: one-pass spacings (the island's
    #                                                                   reduction is their workgroup 0), counts by search
    assert kernels(1500, "multinomial") == "k_f_spacing_onepass<with k_reduce2>+k_ancestors2+k_propagate"
    mv = kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=4)
    ymv = [np.zeros((1, 4)) for _ in range(4)]
    pf = pa.SMC(fk=ssm.Bootstrap(ssm=mv, data=ymv), N=1 << 12, seed=1)
    buf = ctypes.create_string_buffer(256)
    _lib.check(_lib.lib().smc_filter_describe(pf._f, buf, 256))
    assert buf.value.decode() == "k_ancestors<fused>+k_propagate_mv [mv_chunks=1] [diagonal factors]"   # (G = covX = covY = I)


def check_two_level_stepwise(N=2048):
    """Two-level path: the summary row of a step is written by the next launch (or by the flush
    that ends every smc_filter_step call) -- stepping one at a time with reads in between must
    equal one run, for T = 1 included; StopIteration past T."""
    rng = np.random.RandomState(3)
    for T in (1, 2, 7):
        y = [np.array([v]) for v in rng.standard_normal(T)]
        mk = lambda: pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=N, seed=5, ESSrmin=1.0)
        a = mk()
        a.run()
        b = mk()
        for t in range(T):
            next(b)
            assert b.summaries.ESSs[-1] == a.summaries.ESSs[t] and b.logLt == a.summaries.logLts[t]
            assert abs(np.sum(b.W) - 1.0) < 1e-12
        assert a.logLt == b.logLt and np.array_equal(a.X, b.X)
        assert a.summaries.rs_flags == b.summaries.rs_flags == [False] + [True] * (T - 1)
        try:
            next(a)
            raise AssertionError("no StopIteration past T")
        except StopIteration:
            pass


def check_reduce2_wide(monkeypatch, cases, y):
    """Islands of 1025 .. 4096 tiles: the reduction's launch is one workgroup of 1024 threads (k_reduce2w) -- against the
    256-thread kernel (SMC_NO_WIDE keeps it at these sizes) the same run bit for bit: full chunks and a ragged last
    one; the production scheme, a strict filter (its G_b are fractions, not integers), two islands."""
    for n, kw in cases:
        got = {}
        for name, env in (("wide", {}), ("narrow", {"SMC_NO_WIDE": "1"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=n, seed=99, ESSrmin=1.0,
                        resampling="systematic", **kw)
            assert ("k_reduce2w+" in describe(pf)) == (name == "wide"), describe(pf)
            pf.run()
            got[name] = (np.array(pf.A), np.array(pf.X), np.array(pf.wgts.lw), np.array(pf.logLt))
            monkeypatch.undo()
        assert all(np.array_equal(u, v) for u, v in zip(got["wide"], got["narrow"])), (n, kw)


def check_two_level_large(golden, monkeypatch, log2N=21, T=6):
    """More than 1024 tiles per island: k_reduce2 walks the partials in chunks.  Production
    (Philox) mode against the flat-Q62 path on the same counters: the same particle system up to
    near-ties, and the exact-count switch changes nothing."""
    g = golden("kalman_toy")
    y = [np.atleast_1d(v) for v in np.squeeze(g["y"])][:T]
    N = 1 << log2N
    runs = {}
    for name, env in (("two", {}), ("exact", {"SMC_EXACT_COUNTS": "1"}), ("flat", {"SMC_FLAT_CDF": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=N, seed=99, ESSrmin=1.0,
                    resampling="systematic")
        pf.run()
        runs[name] = (np.array(pf.A), np.array(pf.X), list(pf.summaries.logLts), list(pf.summaries.rs_flags))
        monkeypatch.undo()
    assert all(runs["two"][3][1:]) and runs["two"][3] == runs["flat"][3]
    assert np.array_equal(runs["two"][0], runs["exact"][0]) and np.array_equal(runs["two"][1], runs["exact"][1])
    assert rel(runs["two"][2], runs["flat"][2]) < 1e-6       # (two particle systems after the first near-tie)
    check_reduce2_wide(monkeypatch, ((N, {}), (3 * (1 << 20) + 777, {}), (N + 4096 * 3 + 5, {"strict_ancestors": True}),
                                     (1 << 20 | 12345, {"n_islands": 2})), y[:4])
    # the k_reduce2w route against the oracle, every step, every ancestor
    check_oracle_at_size("toy", *MODELS["toy"], N, 4, "systematic", 1.0, replay=False, seed=99)
    ll, _ = orc.kalman_loglik(orc.ToySSM(0.2), y)
    assert abs(runs["two"][2][-1] - ll) < 0.05


def check_graph_replay_matches_direct(golden, N=5000):
    """The hipGraph path (24 steps per graph, slot parity baked into the nodes) against plain
    launches, entered at odd and even time indices and with adaptive resampling."""
    g = golden("kalman_toy")
    y = [np.atleast_1d(v) for v in np.squeeze(g["y"])][:130]
    mk = lambda graph: pa.SMC(fk=ssm.Bootstrap(ssm=kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=1.2),
                                               data=y), N=N, seed=11, use_graph=graph, collect="off")
    a = mk(False)
    a.run()
    b = mk(True)
    for chunk in (7, 53, 26, 1, 43):             # odd entry, even entry, short tails
        b.step_async(chunk)
    assert b.t == 130 and a.logLt == b.logLt and np.array_equal(a.X, b.X)
    s = a._summ()[0]
    assert 0 < s[:, 4].sum() < 129               # both branches of the resample decision
    assert np.array_equal(s, b._summ()[0])


def check_normals_on_host_t(golden, monkeypatch, sizes=(5000, 4096), T=40):
    """k_propagate starts the step's normals on the time index the host passes with the launch
    (FArgs::tk) and falls back to the device record's t: with the feature off (SMC_NO_TK=1) the run
    is the same run -- flat path (N = 5000) and two-level path (N = 4096), adaptive resampling."""
    g = golden("kalman_toy")
    y = [np.atleast_1d(v) for v in np.squeeze(g["y"])][:T]
    for N in sizes:
        runs = []
        for off in (False, True):
            if off:
                monkeypatch.setenv("SMC_NO_TK", "1")
            pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=1.2), data=y),
                        N=N, seed=11, n_islands=2, collect="off")
            pf.step_async(13)
            pf.run()
            runs.append((pf.logLts_islands.copy(), pf._get(_lib.FIELD_X, 1).copy(), pf._summ().copy()))
            monkeypatch.delenv("SMC_NO_TK", raising=False)
        assert all(np.array_equal(u, v) for u, v in zip(*runs)), N
        assert 0 < runs[0][2][0, :, 4].sum() < T - 1


def check_unfused_path(golden, monkeypatch):
    """The k_prepare + k_ancestors<false> path (normally taken beyond 2048 workgroups per
    launch) at test sizes: replay parity, and the same Philox run as the fused path."""
    monkeypatch.setenv("SMC_FORCE_UNFUSED", "1")
    for case in ("toy_systematic", "toy_stratified", "toy_multinomial"):
        check_filter_replay(golden, case, "toy", "bootstrap", T=20)
    check_filter_replay(golden, "toy_systematic", "toy", "bootstrap", T=20, N=4096)
    g = golden("kalman_toy")
    y = [np.atleast_1d(v) for v in np.squeeze(g["y"])][:30]
    mk = lambda: pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=5000, seed=4, n_islands=2,
                        collect="off")
    a = mk()
    a.run()
    monkeypatch.delenv("SMC_FORCE_UNFUSED")
    b = mk()
    b.run()
    # same particles; the evidence is summed in (p, k) pairs on the two-level step and against the
    # maximum on the flat one: equal up to the rounding of a sum of N terms
    assert np.array_equal(a.X, b.X) and np.allclose(a.logLts_islands, b.logLts_islands, rtol=1e-13, atol=0)


def check_merged_reduce_ab(golden, monkeypatch, sizes=(4096, 3000), T=25):
    """Multinomial on the two-level step: the island's reduction as workgroup 0 of the one-pass spacings
    kernel (default) and as a launch of its own (SMC_PATH_SPLIT_REDUCE) are the same run bit for bit --
    adaptive resampling (steps that resample and steps that do not: the draws made side by side with the
    reduction are then dropped), islands, SMC^2's frozen batches."""
    g = golden("kalman_toy")
    y = [np.atleast_1d(v) for v in np.squeeze(g["y"])][:T]
    for N in sizes:
        runs = []
        for split in (False, True):
            if split:
                monkeypatch.setenv("SMC_SPLIT_REDUCE", "1")
            pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=1.5), data=y), N=N,
                        resampling="multinomial", ESSrmin=0.5, seed=21, n_islands=3, collect="off")
            assert ("<with k_reduce2>" in describe(pf)) == (not split), describe(pf)
            pf.step_async(T // 2)
            pf.run()
            flags = pf._summ()[:, :, 4]
            runs.append((np.array(pf.X), np.array(pf.A), pf.logLts_islands.copy(), flags.copy()))
            if split:
                monkeypatch.delenv("SMC_SPLIT_REDUCE")
        assert 0 < runs[0][3][:, 1:].mean() < 1                     # both kinds of step occurred
        for a, b in zip(runs[0], runs[1]):
            assert np.array_equal(a, b)


def check_small_filter_equals_general(golden, monkeypatch, full=True):
    """N <= 1024: the single-launch, single-workgroup filter (smc_filter_small.h) gives the
    bits of the multi-kernel path -- models, schemes, adaptive resampling, islands, stepping
    one step at a time, history -- and replays the reference's run."""
    g = golden("kalman_toy")
    y = [np.atleast_1d(v) for v in np.squeeze(g["y"])][:60 if full else 30]
    cases = [
        (lambda: kalman.ToySSM(0.2), ssm.Bootstrap, "systematic", 1000, 0.5),
        (lambda: kalman.ToySSM(0.2), ssm.Bootstrap, "systematic", 256, 0.5),
        (lambda: kalman.ToySSM(0.2), ssm.Bootstrap, "stratified", 200, 0.8),
        (lambda: kalman.ToySSM(0.2), ssm.Bootstrap, "systematic", 1024, 0.5),
        (lambda: kalman.ToySSM(0.2), ssm.Bootstrap, "stratified", 512, 0.5),
        (lambda: kalman.ToySSM(0.2), ssm.Bootstrap, "stratified", 777, 1.0),
        (lambda: kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=1.5), ssm.Bootstrap, "systematic", 300, 0.5),
        (lambda: kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=0.3), ssm.GuidedPF, "systematic", 640, 0.5),
        (lambda: ssm.StochVol(), ssm.Bootstrap, "systematic", 1000, 0.7),
        (lambda: ssm.Gordon_etal(), ssm.Bootstrap, "stratified", 256, 0.5),
        (lambda: ssm.ThetaLogistic(), ssm.Bootstrap, "systematic", 100, 0.5),
        (lambda: ssm.StochVolLeverage(phi=-0.4), ssm.Bootstrap, "systematic", 1, 0.5),
        (lambda: ssm.DiscreteCox(mu=0.5, sigma=0.4, phi=0.9), ssm.Bootstrap, "systematic", 400, 0.5),
    ]
    if not full:                     # the fiber emulator is slow: a representative subset
        cases = [cases[0], cases[1], cases[5], cases[7], cases[9], cases[11]]
    for mk, cls, scheme, N, essr in cases:
        runs = []
        for small in (True, False):
            if small:
                monkeypatch.delenv("SMC_NO_SMALL", raising=False)
            else:
                monkeypatch.setenv("SMC_NO_SMALL", "1")
            pf = pa.SMC(fk=cls(ssm=mk(), data=y), N=N, resampling=scheme, ESSrmin=essr, seed=5,
                        n_islands=3, store_history=(N == 300))
            pf.step_async(7)
            for _ in range(4):
                next(pf)
            pf.run()
            runs.append((pf.logLts_islands.copy(), pf._get(_lib.FIELD_X, 2).copy(), pf._summ().copy(),
                         pf._get(_lib.FIELD_A, 1).copy(), pf.wgts.lw.copy(),
                         pf.hist.X[20].copy() if N == 300 else None))
        for u, v in zip(*runs):
            assert (u is None and v is None) or np.array_equal(u, v, equal_nan=True), (scheme, N)
    monkeypatch.delenv("SMC_NO_SMALL", raising=False)
    # and the reference's own run through it (replay of its draws)
    if full:
        check_filter_replay(golden, "toy_systematic", "toy", "bootstrap", T=40)
        check_filter_replay(golden, "toy_stratified", "toy", "bootstrap", T=40)
        check_filter_replay(golden, "toy_multinomial", "toy", "bootstrap", T=40)
        check_filter_replay(golden, "lg_adaptive", "lg_adaptive", "bootstrap", T=60)
        for case in ("toy_systematic", "toy_stratified", "toy_multinomial"):      # one-wave variant
            check_filter_replay(golden, case, "toy", "bootstrap", T=30, N=256)
            check_filter_replay(golden, case, "toy", "bootstrap", T=30, N=100)


def check_edge_sizes():
    """Ragged and tiny populations, single-step runs, every scheme (partial
    wavefronts, partial tiles, tiles with no offspring)."""
    rng = np.random.RandomState(0)
    y = [np.array([v]) for v in rng.standard_normal(6)]
    for N in (1, 2, 3, 63, 64, 65, 255, 257, 1023, 1025, 2049):
        for scheme in ("systematic", "stratified", "multinomial"):
            for T in (1, 6):
                pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.5), data=y[:T]), N=N,
                            resampling=scheme, seed=3, ESSrmin=1.0)
                pf.run()
                assert np.isfinite(pf.logLt), (N, scheme, T)
                assert pf.X.shape == (N,) and np.all(np.isfinite(pf.X))
                assert abs(pf.W.sum() - 1.0) < 1e-12
                if T > 1:
                    A = pf.A
                    assert A.min() >= 0 and A.max() < N and np.all(np.diff(A) >= 0)


def check_device_history(golden):
    """store_history=True on the fused path: the history stays in HBM
    (keep_history) and hist.X / hist.A / hist.wgts / compute_trajectories are
    served from it -- against the reference's own history (tests/golden/history.npz,
    smoothing.py:181-219), replaying its draws."""
    g = golden("history")
    mk_dev, mk_orc = MODELS["lg_adaptive"]
    N, T = int(g["N"]), int(g["T"])
    y = list(g["y"])
    np.random.seed(int(g["run_seed"]))
    rec = orc.RecordingRNG()
    o = orc.run_filter(mk_orc(), y, N, "systematic", 0.5, rng=rec)
    assert o["final_logLt"] == float(g["logLt"])
    z, u = tapes_from_oracle(rec.tape, T, N, "systematic")
    pf = pa.SMC(fk=ssm.Bootstrap(ssm=mk_dev(), data=y), N=N, resampling="systematic",
                ESSrmin=0.5, replay=(z, u), store_history=True)
    assert not pf._needs_per_step_host()             # one asynchronous launch sequence
    pf.run()
    h = pf.hist
    assert h.T == T and h.N == N and len(h.X) == T and h.A[0] is None
    assert pf.summaries.rs_flags == [bool(v) for v in g["rs_flags"]]
    for t in range(T):
        assert np.array_equal(h.X[t], g["hist_X"][t])                 # IEEE + - * / only
        assert np.array_equal(h.wgts[t].lw, g["hist_lw"][t])
        assert rel(h.wgts[t].W, g["hist_W"][t]) < 1e-10
        if t:
            assert np.array_equal(h.A[t], g["hist_A"][t - 1])
    assert np.array_equal(h.X[-1], pf.X) and np.array_equal(h.A[-1], pf.A)
    assert np.array_equal(h.compute_trajectories(), g["trajectories"])
    # extract_one_trajectory (smoothing.py:256-269): same draw, same line as from the fixture
    np.random.seed(99)
    traj = h.extract_one_trajectory()
    np.random.seed(99)
    n = int(np.searchsorted(np.cumsum(g["hist_W"][-1]), np.random.rand()))
    want = []
    for t in reversed(range(T)):
        if t < T - 1:
            n = g["hist_A"][t][n]                # hist_A[t] = A of step t+1
        want.append(g["hist_X"][t][n])
    assert len(traj) == T and np.array_equal(np.array(traj), np.array(want[::-1]))
    import pytest
    with pytest.raises(IndexError):
        h.X[T]
    # a filter created without history refuses
    pf2 = pa.SMC(fk=ssm.Bootstrap(ssm=mk_dev(), data=y), N=N, seed=3)
    pf2.run()
    with pytest.raises(Exception):
        pf2._history(_lib_field_x(), 0)


def _lib_field_x():
    from particles_amd import _lib
    return _lib.FIELD_X


def check_partial_history(N=3000, T=23):
    """``store_history=<callable>`` (PartialParticleHistory, smoothing.py:164-184) on the fused path:
    the run synchronises at the save times only; what it saves is what a run with the whole history
    holds at those times (same seed: same run), keys as in the reference (the saved t)."""
    rng = np.random.RandomState(2)
    y = [np.array([v]) for v in np.cumsum(rng.standard_normal(T))]
    fk = lambda: ssm.Bootstrap(ssm=kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=1.2), data=y)
    save = lambda t: t % 7 == 3 or t == T - 1
    full = pa.SMC(fk=fk(), N=N, seed=5, store_history=True)
    full.run()
    calls = []
    part = pa.SMC(fk=fk(), N=N, seed=5, store_history=lambda t: (calls.append(t), save(t))[1])
    part.run()
    assert part._fused and sorted(part.hist.X) == [t for t in range(T) if save(t)] == sorted(part.hist.wgts)
    for t in part.hist.X:
        assert np.array_equal(part.hist.X[t], full.hist.X[t])
        assert np.array_equal(part.hist.wgts[t].lw, full.hist.wgts[t].lw)
        assert abs(part.hist.wgts[t].ESS - full.hist.wgts[t].ESS) < 1e-9 * N
    assert part.logLt == full.logLt and part.summaries.logLts == full.summaries.logLts
    assert part.summaries.rs_flags == full.summaries.rs_flags
    # stepping by hand keeps the reference's per-step save
    byhand = pa.SMC(fk=fk(), N=N, seed=5, store_history=save)
    for _ in byhand:
        pass
    assert sorted(byhand.hist.X) == sorted(part.hist.X)
    assert all(np.array_equal(byhand.hist.X[t], part.hist.X[t]) for t in part.hist.X)


def check_rolling_history(N=3000, T=23, ks=(2, 5, 9)):
    """store_history=k (RollingParticleHistory, smoothing.py:186-219) on the fused path: a ring of
    k slots in HBM.  Same run as with the whole history resident -- final state, summaries -- and
    the window holds exactly its k most recent steps (X, A, log-weights, W, genealogy); older
    steps are refused.  Both CDF paths (N a power of two or not), stepping with reads in between."""
    rng = np.random.RandomState(8)
    y = [np.array([v]) for v in 0.5 * np.cumsum(rng.standard_normal(T))]
    for n in (N, 4096):
        mk = lambda h: pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.3), data=y), N=n, seed=12, ESSrmin=0.7,
                              store_history=h)
        full = mk(True)
        full.run()
        Bfull = full.hist.compute_trajectories()
        for k in ks:
            r = mk(k)
            assert isinstance(r.hist, pa.collectors.DeviceRollingParticleHistory)
            for t in range(T):
                next(r)
                assert r.hist.T == min(t + 1, k) and len(r.hist.X) == r.hist.T
                if t in (0, 1, k, T // 2):
                    assert np.array_equal(r.hist.X[-1], full.hist.X[t])
                    assert np.array_equal(r.hist.X[0], full.hist.X[max(0, t + 1 - k)])
            assert np.array_equal(r.X, full.X) and np.array_equal(r._summ(), full._summ())
            for i in range(k):
                t = T - k + i
                assert np.array_equal(r.hist.X[i], full.hist.X[t])
                assert np.array_equal(r.hist.wgts[i].lw, full.hist.wgts[t].lw)
                assert np.array_equal(r.hist.wgts[i].W, full.hist.wgts[t].W)
                a_r, a_f = r.hist.A[i], full.hist.A[t]
                assert (a_r is None) == (a_f is None) and (a_r is None or np.array_equal(a_r, a_f))
            assert np.array_equal(r.hist.compute_trajectories(), Bfull[T - k:])
            try:
                r._history(_lib.FIELD_X, T - k - 1)
                raise AssertionError("a step outside the window was served")
            except RuntimeError:
                pass
    # a window as long as the run is the whole history
    w = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.3), data=y), N=N, seed=12, store_history=T + 5)
    w.run()
    assert len(w.hist.X) == T


def check_device_history_philox(N, T, golden):
    """Production mode: history on/off give the same run; the genealogy obeys
    B_{t-1} = A_t[B_t] (smoothing.py:213-216), islands > 1 included."""
    g = golden("kalman_toy")
    y = [np.atleast_1d(v) for v in np.squeeze(g["y"])][:T]
    mk = lambda **kw: pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=N, seed=77, **kw)
    a, b = mk(), mk(store_history=True, n_islands=1)
    a.run(); b.run()
    assert a.logLt == b.logLt and np.array_equal(a.X, b.X) and np.array_equal(a.A, b.A)
    B = b.hist.compute_trajectories()
    assert B.shape == (T, N) and np.array_equal(B[-1], np.arange(N))
    for t in (T - 1, T // 2, 1):
        assert np.array_equal(B[t - 1], b.hist.A[t][B[t]])
    assert np.all(np.diff(B[0]) >= 0)                  # ancestors stay sorted (systematic)
    assert np.array_equal(b.hist.X[3], mk_step_X(mk, 4))


def mk_step_X(mk, nsteps):
    p = mk()
    p.step_async(nsteps)
    return p.X


def check_filter_stepwise(golden):
    """next(pf) one step at a time == run(), and the iterator protocol."""
    g = golden("toy_systematic")
    y = list(g["y"])[:12]
    a = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=3000, seed=5)
    a.run()
    b = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=3000, seed=5)
    n = 0
    for _ in b:
        n += 1
    assert n == 12 and b.t == 12
    import pytest
    with pytest.raises(StopIteration):
        next(b)
    assert a.logLt == b.logLt and np.array_equal(a.X, b.X)
    assert a.summaries.logLts == b.summaries.logLts and len(a.summaries.ESSs) == 12


def check_filter_philox_vs_c(N, T, golden, sigmaY=0.2):
    """Production (Philox) mode against the C oracle running the same counter
    stream: integer pipeline identical, Gaussians equal to ~1 ulp.  A small
    sigmaY makes the weights degenerate (few parents, many offspring each)."""
    g = golden("kalman_toy")
    y = np.ascontiguousarray(np.squeeze(g["y"]))[:T]
    summ = np.zeros(4 * T)
    ll_c = orc.clib().orc_toy_filter_philox(orc._dp(y), T, N, 1.0, 1.0, sigmaY, 1.0, 0.5, 2024,
                                            orc._dp(summ))
    pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(sigmaY), data=[np.array([v]) for v in y]),
                N=N, seed=2024)
    pf.run()
    s = summ.reshape(T, 4)
    assert pf.summaries.rs_flags == [bool(v) for v in s[:, 3]]
    assert rel(pf.summaries.ESSs, s[:, 0]) < 1e-7
    assert abs(pf.logLt / ll_c - 1) < 1e-9


def check_filter_kalman(N, golden, scheme="systematic"):
    g = golden("kalman_toy")
    y = [np.atleast_1d(v) for v in np.squeeze(g["y"])]
    lls = []
    for s in range(4):
        pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=N, seed=100 + s,
                    resampling=scheme)
        pf.run()
        lls.append(pf.logLt)
    lls = np.array(lls)
    tol = 8.0 / np.sqrt(N) * np.sqrt(len(y)) + 0.05
    assert abs(lls.mean() - float(g["loglik"])) < tol, (lls, float(g["loglik"]))


def check_islands(N, T, golden, scheme="stratified"):
    """Island k of a batched filter == a single filter with island_offset=k."""
    g = golden("kalman_toy")
    y = [np.atleast_1d(v) for v in np.squeeze(g["y"])][:T]
    fk = ssm.Bootstrap(ssm=kalman.LinearGauss(rho=0.95, sigmaX=0.7, sigmaY=0.5), data=y)
    pf = pa.SMC(fk=fk, N=N, seed=31, n_islands=3, resampling=scheme, collect="off", ESSrmin=0.7)
    pf.run()
    ll = pf.logLts_islands
    assert len(set(ll.tolist())) == 3
    for k in (0, 2):
        one = pa.SMC(fk=fk, N=N, seed=31, resampling=scheme, collect="off", ESSrmin=0.7,
                     island_offset=k)
        one.run()
        assert one.logLt == ll[k]
        assert np.array_equal(one.X, pf._get(_lib.FIELD_X, k))
    out = pa.multiSMC(fk=fk, N=N, nruns=3, out_func=lambda p: p.logLt, resampling=scheme)
    assert [d["run"] for d in out] == [0, 1, 2] and all(np.isfinite(d["output"]) for d in out)


def check_permute_islands(N, golden, tol=0.3, T=40, t0=15):
    """One model per island (SMC^2: one theta each) and theta-level resampling of whole
    filters (smc_samplers.py:319-361): identity and round trips are exact, copies carry
    particles, evidence and parameters, and every island goes on under its new theta."""
    g = golden("kalman_toy")
    y = [np.atleast_1d(v) for v in np.squeeze(g["y"])][:T]
    sig = [0.2, 0.5, 1.5]
    mk = lambda: pa.SMC(fk=[ssm.Bootstrap(ssm=kalman.ToySSM(s), data=y) for s in sig], N=N, seed=77,
                        collect="off")
    ref = mk()
    ref.run()
    assert ref.n_islands == 3 and len(set(ref.logLts_islands.tolist())) == 3
    # the per-island parameters are really used: evidence against the exact Kalman one
    for k, s_ in enumerate(sig):
        ll, _ = orc.kalman_loglik(orc.ToySSM(s_), y)
        assert abs(ref.logLts_islands[k] - ll) < tol, (k, ref.logLts_islands[k], ll)   # MC error + log-bias
    # identity, and a permutation undone before the next step: bit-identical runs
    a = mk()
    a.step_async(t0)
    a.permute_islands([0, 1, 2])
    a.permute_islands([1, 2, 0])
    a.permute_islands([2, 0, 1])
    a.step_async(T - t0)
    assert np.array_equal(a.logLts_islands, ref.logLts_islands)
    assert np.array_equal(a._get(_lib.FIELD_X, 1), ref._get(_lib.FIELD_X, 1))
    # copies: island 2's filter takes over slots 0 and 1, island 0's goes to slot 2
    b = mk()
    b.step_async(t0)
    before = b.logLts_islands.copy()
    x2 = b._get(_lib.FIELD_X, 2).copy()
    b.permute_islands([2, 2, 0])
    assert np.array_equal(b.logLts_islands, before[[2, 2, 0]])
    assert np.array_equal(b._get(_lib.FIELD_X, 0), x2) and np.array_equal(b._get(_lib.FIELD_X, 1), x2)
    import pytest
    with pytest.raises(Exception):
        b._get(_lib.FIELD_A, 0)                  # undefined right after the permutation
    b.step_async(T - t0)
    ll = b.logLts_islands
    assert ll[0] != ll[1]                        # same state, different random streams from here on
    for k, src in enumerate([2, 2, 0]):          # and each goes on under the theta it inherited
        full, _ = orc.kalman_loglik(orc.ToySSM(sig[src]), y)
        head, _ = orc.kalman_loglik(orc.ToySSM(sig[src]), y[:t0])
        assert abs((ll[k] - before[src]) - (full - head)) < max(0.3, tol), (k, ll[k] - before[src], full - head)
    with pytest.raises(ValueError):
        b.permute_islands([0, 1])
    # PMCMC move: islands accepted from a second batch run on other thetas
    sig2 = [0.3, 0.6, 1.0]
    cur, prop = mk(), pa.SMC(fk=[ssm.Bootstrap(ssm=kalman.ToySSM(s), data=y) for s in sig2], N=N,
                             seed=78, collect="off")
    cur.step_async(t0)
    prop.step_async(t0)
    lc, lp = cur.logLts_islands.copy(), prop.logLts_islands.copy()
    xp1 = prop._get(_lib.FIELD_X, 1).copy()
    cur.accept_islands_from(prop, [False, True, False])
    assert np.array_equal(cur.logLts_islands, [lc[0], lp[1], lc[2]])
    assert np.array_equal(cur._get(_lib.FIELD_X, 1), xp1)
    cur.step_async(T - t0)
    full, _ = orc.kalman_loglik(orc.ToySSM(sig2[1]), y)
    head, _ = orc.kalman_loglik(orc.ToySSM(sig2[1]), y[:t0])
    assert abs((cur.logLts_islands[1] - lp[1]) - (full - head)) < 0.3      # goes on under the new theta
    short = mk()
    short.step_async(3)
    with pytest.raises(Exception):
        cur.accept_islands_from(short, [True, True, True])                  # different time index


def check_mv_kalman(N, d, fk, scheme="systematic"):
    """Production (Philox) mode of the multivariate filter against the exact
    Kalman likelihood (kalman.py:483-505 restated in the oracle)."""
    rng = np.random.RandomState(5)
    T = 10
    om = orc.Guarniero(alpha=0.4, dx=d)
    x = np.zeros(d)
    y = []
    for t in range(T):
        x = (om.F @ x if t else np.zeros(d)) + rng.standard_normal(d)
        y.append((x + rng.standard_normal(d)).reshape(1, d))
    ll, _ = orc.kalman_loglik(om, y)
    cls = ssm.Bootstrap if fk == "bootstrap" else ssm.GuidedPF
    lls = []
    for s in range(3):
        pf = pa.SMC(fk=cls(ssm=kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=d), data=y), N=N,
                    seed=10 + s, resampling=scheme)
        pf.run()
        assert pf.X.shape == (N, d) and np.isfinite(pf.logLt)
        lls.append(pf.logLt)
    tol = (0.05 if fk == "guided" else 4.0 * d / np.sqrt(N) * 8) + 0.02
    assert abs(np.mean(lls) - ll) < tol, (lls, ll)


def check_mv_collapsed(N, d, T=6):
    """MVLinearGauss guided filter with the collapsed form of the optimal proposal's weight
    (SMC_FLAG_COLLAPSED_PROPOSAL: log G = log p(y_t | x_{t-1})): the particles are the default
    path's bit for bit as long as the resampling decisions agree, the log-weights equal up to
    the rounding of the three-term expression it replaces (state_space_models.py:380-392), the
    evidence agrees with the exact Kalman likelihood (kalman.py:483-505)."""
    rng = np.random.RandomState(5)
    om = orc.Guarniero(alpha=0.4, dx=d)
    x = np.zeros(d)
    y = []
    for t in range(T):
        x = (om.F @ x if t else np.zeros(d)) + rng.standard_normal(d)
        y.append((x + rng.standard_normal(d)).reshape(1, d))
    ll, _ = orc.kalman_loglik(om, y)
    mk = lambda c, ess: pa.SMC(fk=ssm.GuidedPF(ssm=kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=d), data=y),
                               N=N, seed=31, ESSrmin=ess, collapsed_proposal=c)
    # ESSrmin = 0: never resample -> identical particle systems, the weights can be compared
    a, b = mk(False, 0.0), mk(True, 0.0)
    a.run()
    b.run()
    assert "collapsed" in describe(b) and "collapsed" not in describe(a)
    assert np.array_equal(a.X, b.X)
    assert np.allclose(a.wgts.lw, b.wgts.lw, rtol=0, atol=2e-10 * T * d), np.max(np.abs(a.wgts.lw - b.wgts.lw))
    assert abs(a.logLt - b.logLt) < 1e-9 * abs(a.logLt)
    c, c0 = mk(True, 0.5), mk(False, 0.5)
    c.run()
    c0.run()
    assert c.summaries.rs_flags == c0.summaries.rs_flags and abs(c.logLt - c0.logLt) < 1e-8 * abs(c0.logLt)
    if N >= 1 << 14:
        assert abs(c.logLt - ll) < 0.07 * max(1.0, np.sqrt((1 << 17) / N)), (c.logLt, ll)
    with np.errstate(all="ignore"):        # the flag is the guided MV filter's: ignored elsewhere
        e = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=d), data=y), N=N,
                   seed=31, collapsed_proposal=True)
        e.run()
        assert "collapsed" not in describe(e) and np.isfinite(e.logLt)


def check_mv_diag_equals_dense(monkeypatch, cases=((1500, 32), (700, 20), (900, 4), (600, 16)), T=4):
    """MVLinearGauss with diagonal G / covX / covY / cov0 (every Guarniero model; BASELINE C4): k_propagate_mv applies the
    step's triangular factors element by element (smc_filter_mv.h "DG") -- the SAME run, bit for bit, as the dense MFMA
    products (SMC_MV_DENSE=1): guided, bootstrap, collapsed weight, APF; Philox and replayed draws; d = DP and d < DP;
    and a model whose matrices are NOT diagonal never takes the element-wise form."""
    rng = np.random.RandomState(8)
    for N, d in cases:
        y = [rng.standard_normal((1, d)) for _ in range(T)]
        mod = lambda: kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=d)
        z = np.random.RandomState(N).standard_normal((T, 1, N, d))
        u = np.random.RandomState(N + 1).rand(T, 1, 1)
        variants = [("guided", dict(fk=ssm.GuidedPF(ssm=mod(), data=y))),
                    ("bootstrap", dict(fk=ssm.Bootstrap(ssm=mod(), data=y))),
                    ("collapsed", dict(fk=ssm.GuidedPF(ssm=mod(), data=y), collapsed_proposal=True)),
                    ("apf", dict(fk=ssm.AuxiliaryPF(ssm=mod(), data=y))),
                    ("guided-replay", dict(fk=ssm.GuidedPF(ssm=mod(), data=y), replay=(z, u)))]
        for name, kw in variants:
            runs = []
            # (the same workgroups for both forms: the host picks 8 chunks of 256 particles per workgroup for the
            #  element-wise kernel and 4 for the dense one at N = 2^20 -- other partials, other roundings of their sum)
            monkeypatch.setenv("SMC_MV_CHUNKS", "8" if N >= 1 << 20 else ("2" if N >= 1 << 17 else "1"))
            for dense in (False, True):
                if dense:
                    monkeypatch.setenv("SMC_MV_DENSE", "1")
                pf = pa.SMC(N=N, seed=12, ESSrmin=1.0, store_history=True, collect="off", **kw)
                assert ("[diagonal factors]" in describe(pf)) == (not dense), describe(pf)
                pf.run()
                runs.append(pf)
                monkeypatch.delenv("SMC_MV_DENSE", raising=False)
            a, b = runs
            assert np.any(a._summ()[0, 1:, 4] != 0), name
            assert np.array_equal(a._summ(), b._summ()), (N, d, name)
            for t in range(T):
                assert np.array_equal(a._history(_lib.FIELD_X, t), b._history(_lib.FIELD_X, t)), (N, d, name, t)
                assert np.array_equal(a._history(_lib.FIELD_LW, t), b._history(_lib.FIELD_LW, t)), (N, d, name, t)
            assert a.logLt == b.logLt
            monkeypatch.delenv("SMC_MV_CHUNKS", raising=False)
    y = [rng.standard_normal((1, 6)) for _ in range(3)]
    pf = pa.SMC(fk=ssm.GuidedPF(ssm=MODELS["mvd8"][0](), data=y), N=500, seed=1)
    assert "[diagonal factors]" not in describe(pf) and "k_propagate_mv" in describe(pf)


def check_smc2(Ntheta=64, Nx=128, T=30, seed=3, big_Nx=(), big_N=16, big_T=12):
    """SMC^2 with the theta level on the device (particles_amd.smc2, smc_samplers.py:1038-1167).
    (1) With the theta-level ESS threshold at 0 nothing ever stops: the device's theta weights
    must then BE the islands' log-evidences, whatever the number of steps enqueued per sync.
    (2) A full run: resample-move events happen, are dealt with at the right step (a frozen batch
    does no step beyond the stop), the filters of all thetas stay in lock step, the posterior of
    the unknown parameter covers the value the data were simulated with, the evidence of the
    whole model is finite; (3) the exchange step doubles N_x and keeps going."""
    from particles_amd import smc2
    rng = np.random.RandomState(seed)
    sig = 0.3
    x = np.cumsum(rng.standard_normal(T))
    y = [np.array([v]) for v in x + sig * rng.standard_normal(T)]
    prior = smc2.IndepPrior(sigmaY=("lognormal", np.log(0.5), 0.5))
    mk = lambda **kw: smc2.SMC2(ssm_cls=lambda sigmaY: kalman.LinearGauss(rho=1.0, sigmaX=1.0, sigmaY=sigmaY,
                                                                          sigma0=1.0),
                                prior=prior, data=y, init_Nx=Nx, N=Ntheta, seed=seed, **kw)
    # (1)
    a = mk(ESSrmin=0.0, sync_every=7)
    a.run()
    assert a.t == T and not a.move_times
    assert np.array_equal(a.lw, a.pf.logLts_islands)
    b = mk(ESSrmin=0.0, sync_every=T)
    b.run()
    assert np.array_equal(a.lw, b.lw) and a.logLt == b.logLt
    # the theta-level ESS the device logged is the ESS of those weights
    s = a.pf._summ()                                               # (Ntheta, T, 5)
    lwt = np.cumsum(s[:, :, 2], axis=1)
    for t in (0, T // 2, T - 1):
        w = np.exp(lwt[:, t] - lwt[:, t].max())
        assert abs(a.ESSs[t] / (w.sum() ** 2 / np.sum(w ** 2)) - 1) < 1e-12
    # (2)
    c = mk(ESSrmin=0.5, sync_every=5, nmcmc=2)
    c.run()
    assert c.t == T and len(c.move_times) >= 1 and np.isfinite(c.logLt)
    assert len(c.ESSs) == T and c.pf.t == T
    m, sd = c.posterior_mean()["sigmaY"], c.posterior_sd()["sigmaY"]
    assert 0.05 < m < 1.5 and abs(m - sig) < 4 * sd + 0.15, (m, sd)
    assert all(0.0 <= r <= 1.0 for r in c.acc_rates) and len(c.acc_rates) == 2 * len(c.move_times)
    # every island of the surviving batch is at step T with a finite evidence
    assert np.all(np.isfinite(c.pf.logLts_islands))
    # (2b) the same on the multi-kernel paths (N_x = 2048: two-level CDF; 3000: flat CDF)
    if big_Nx:
        mkb = lambda nx, **kw: smc2.SMC2(
            ssm_cls=lambda sigmaY: kalman.LinearGauss(rho=1.0, sigmaX=1.0, sigmaY=sigmaY, sigma0=1.0),
            prior=prior, data=y[:big_T], init_Nx=nx, N=min(Ntheta, big_N), seed=seed, **kw)
        for nx in big_Nx:
            d = mkb(nx, ESSrmin=0.99, sync_every=5, nmcmc=1)
            d.run()
            assert d.t == big_T and len(d.ESSs) == big_T and np.isfinite(d.logLt), (d.t, d.ESSs, d.logLt)
            assert len(d.move_times) >= 1, d.ESSs
            d0 = mkb(nx, ESSrmin=0.0, sync_every=big_T)
            d0.run()
            assert np.array_equal(d0.lw, d0.pf.logLts_islands)
    # (3) exchange step: every move is "rejected too often" -> N_x doubles (once: max_Nx)
    e = mk(ESSrmin=0.5, sync_every=4, nmcmc=1, ar_to_increase_Nx=1.01, max_Nx=2 * Nx)
    e.run()
    assert e.t == T and e.Nx == 2 * Nx and e.pf.N == 2 * Nx and np.isfinite(e.logLt)
    assert abs(e.posterior_mean()["sigmaY"] - sig) < 0.3


def check_smc2_vs_reference(golden, R=24, sharded=False, tol_se=3.0, wastefree=False):
    """SMC^2 pinned to the REFERENCE's SMC^2 (smc_samplers.py:1038-1167): tests/golden/smc2_ref.npz holds
    24 independent runs of particles.SMC(fk=SMC2(kalman.LinearGauss, StructDist{rho ~ U(0.3, 0.99),
    sigmaY ~ Gamma(2, 4)}, init_Nx=64, len_chain=4, wastefree=False), N=64) on one simulated data set,
    run by run (tests/golden/make_golden.py smc2_ref).  The device class runs the same algorithm --
    same prior, N_theta, N_x, ESS threshold, 3 random-walk Metropolis steps per move calibrated on the
    weighted theta-particles -- R times with different seeds; every summary must be a draw from the same
    distribution: means within `tol_se` standard errors (both sides' Monte Carlo error counted) for the
    evidence of the whole model, the posterior mean and sd of each parameter, the ESS trajectory and the
    number of resample-move events.  A wrong weight update, move or exchange shifts these by many SEs."""
    from particles_amd import smc2
    # wastefree: fixture smc2_wf_ref = 32 runs of the reference's DEFAULT move (smc_samplers.py:669-684: N = 32
    # chains of len_chain = 4 states, all kept: 128 theta-particles) against the device's waste-free move
    g = golden("smc2_wf_ref" if wastefree else "smc2_ref")
    y = [np.atleast_1d(v) for v in np.squeeze(g["y"])]
    T, N, Nx = int(g["T"]), int(g["N"]), int(g["Nx"])
    prior = smc2.IndepPrior(rho=("uniform", float(g["prior_rho"][0]), float(g["prior_rho"][1])),
                            sigmaY=("gamma", float(g["prior_sigmaY"][0]), float(g["prior_sigmaY"][1])))
    cls = smc2.ShardedSMC2 if sharded else smc2.SMC2
    rec = {k: [] for k in ("logLt", "m_rho", "m_sigmaY", "s_rho", "s_sigmaY", "ESSs", "nmoves")}
    for r in range(R):
        alg = cls(ssm_cls=lambda rho, sigmaY: kalman.LinearGauss(sigmaX=float(g["sigmaX"]), sigmaY=sigmaY, rho=rho),
                  prior=prior, data=y, init_Nx=Nx, N=N, ESSrmin=float(g["ESSrmin"]), nmcmc=int(g["len_chain"]) - 1,
                  seed=500 + r, sync_every=8, **(dict(wastefree=True, len_chain=int(g["len_chain"])) if wastefree else {}))
        assert alg.N == N * (int(g["len_chain"]) if wastefree else 1) and alg.pf.n_islands == alg.N
        alg.run()
        assert alg.t == T and len(alg.ESSs) == T
        m, sd = alg.posterior_mean(), alg.posterior_sd()
        rec["logLt"].append(alg.logLt)
        for k in ("rho", "sigmaY"):
            rec["m_" + k].append(m[k])
            rec["s_" + k].append(sd[k])
        rec["ESSs"].append(alg.ESSs)
        rec["nmoves"].append(len(alg.move_times))
    ref = {k: np.asarray(g[k], dtype=float) for k in ("logLt", "m_rho", "m_sigmaY", "s_rho", "s_sigmaY", "ESSs")}
    ref["nmoves"] = np.asarray(g["rs_flags"], dtype=float).sum(axis=1)
    report = {}

    def same_mean(name, a, b):
        a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
        se = np.sqrt(a.var(ddof=1) / a.size + b.var(ddof=1) / b.size)
        z = (a.mean() - b.mean()) / max(se, 1e-300)
        report[name] = (round(float(a.mean()), 4), round(float(b.mean()), 4), round(float(z), 2))
        return abs(z) <= tol_se

    ok = [same_mean(k, rec[k], ref[k]) for k in ("logLt", "m_rho", "m_sigmaY", "s_rho", "s_sigmaY", "nmoves")]
    E, Er = np.asarray(rec["ESSs"]), ref["ESSs"]                        # (R, T) ESS after every step
    ok.append(same_mean("ESS_mean", E.mean(axis=1), Er.mean(axis=1)))
    for t in (1, T // 4, T // 2, T - 1):
        ok.append(same_mean("ESS_t%d" % t, E[:, t], Er[:, t]))
    print("smc2 vs reference (device mean, reference mean, z):", report)
    # 11 comparisons at 3 SE: a correct implementation fails one of them with probability ~3 %
    # -> allow ONE excursion up to 4 SE, none beyond
    zs = np.array([abs(v[2]) for v in report.values()])
    assert np.all(zs <= tol_se + 1.0) and np.sum(zs > tol_se) <= 1, report
    return report


def check_smc2_wastefree_functional():
    from particles_amd import smc2
    rng = np.random.RandomState(4)
    T = 20
    x = np.cumsum(rng.standard_normal(T)) * 0.5
    y = [np.array([v]) for v in x + 0.3 * rng.standard_normal(T)]
    prior = smc2.IndepPrior(sigmaY=("lognormal", np.log(0.5), 1.0))
    kw = dict(ssm_cls=lambda sigmaY: kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=sigmaY), prior=prior, data=y,
              init_Nx=64, ESSrmin=0.5)
    runs = []
    for seed in (5, 6, 7):
        a = smc2.SMC2(wastefree=True, len_chain=5, N=8, seed=seed, **kw)
        assert a.N == 40 and a.M == 8 and a.P == 5 and a.pf.n_islands == 40 and a.nmcmc == 4
        a.run()
        assert a.t == T and len(a.ESSs) == T and len(a.move_times) >= 1 and a.pf.n_islands == 40
        assert len(a.acc_rates) == 4 * len(a.move_times) and all(len(v) == 40 for v in a.theta.values())
        assert np.isfinite(a.logLt) and np.all(np.isfinite(a.lw))
        runs.append((a.logLt, a.posterior_mean()["sigmaY"]))
    b = smc2.SMC2(nmcmc=4, N=40, seed=5, **kw)
    b.run()
    ll = np.array([r[0] for r in runs])
    assert abs(ll.mean() - b.logLt) < 1.0 and abs(np.mean([r[1] for r in runs]) - b.posterior_mean()["sigmaY"]) < 0.15
    # a filter taken over by a fresh one continues bit for bit (fast-forward + pack / unpack)
    src = pa.SMC(fk=[ssm.Bootstrap(ssm=kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=s_), data=y) for s_ in (0.2, 0.3, 0.5)],
                 N=200, seed=3, collect="off")
    src.step_async(7)
    dst = pa.SMC(fk=[ssm.Bootstrap(ssm=kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=0.9), data=y) for _ in range(3)],
                 N=200, seed=3, collect="off")
    dst.take_islands_from(src, [2, 0, 1])
    assert dst.t == 7
    src.permute_islands(np.array([2, 0, 1]))
    src.run()
    dst.run()
    assert np.array_equal(src.logLts_islands, dst.logLts_islands) and np.array_equal(np.asarray(src.X), np.asarray(dst.X))


def check_generic_path(golden):
    """A user-defined FeynmanKac in Python: template method with device ops."""
    g = golden("kalman_toy")
    y = np.squeeze(g["y"])[:30]

    class MyFK(pa.FeynmanKac):
        def M0(self, N):
            return dists.Normal().rvs(size=N)

        def M(self, t, xp):
            return dists.Normal(loc=xp).rvs(size=xp.shape[0])

        def logG(self, t, xp, x):
            return dists.Normal(loc=x, scale=0.2).logpdf(y[t])

    pa.seed(3)
    np.random.seed(3)
    from particles_amd.collectors import Moments
    pf = pa.SMC(fk=MyFK(T=30), N=4000, collect=[Moments()], store_history=True)
    pf.run()
    ll, means = orc.kalman_loglik(orc.ToySSM(0.2), [np.atleast_1d(v) for v in y])
    assert abs(pf.logLt - ll) < 1.5
    est = np.array([m["mean"] for m in pf.summaries.moments])
    assert np.max(np.abs(est - means)) < 0.15
    assert len(pf.hist.X) == 30 and pf.hist.compute_trajectories().shape == (30, 4000)


def check_apf_and_guided_generic(golden):
    """The template-method step with device operators, on the reference's own draws
    (same numpy seed): the auxiliary particle filter (core.py:299-313; Pitt & Shephard's
    proposal for StochVol, state_space_models.py:475-498) and the guided filter of the same
    model, against the reference's outputs."""
    class GenericSV(ssm.StochVol):                   # keeps the model off the fused loop
        def _device_params(self, fk_kind):
            return None

    for case, cls in (("sv_apf", ssm.AuxiliaryPF), ("sv_guided", ssm.GuidedPF)):
        g = golden(case)
        y = list(g["y"])
        np.random.seed(int(g["run_seed"]))
        pf = pa.SMC(fk=cls(ssm=GenericSV(), data=y), N=int(g["N"]),
                    resampling=str(g["scheme"]), ESSrmin=float(g["ESSrmin"]))
        assert not pf._fused and pf.fk.isAPF == (case == "sv_apf")
        pf.run()
        assert pf.summaries.rs_flags == [bool(v) for v in g["rs_flags"]]
        assert rel(pf.summaries.ESSs, g["ESSs"]) < 1e-9
        assert rel(pf.summaries.logLts, g["logLts"]) < 1e-9
        assert abs(pf.logLt / float(g["logLt"]) - 1) < 1e-10
        same = np.mean(pf.A == g["A"])
        assert same >= 0.999
        if same == 1.0:
            assert np.max(np.abs(pf.X - g["X"])) < 1e-12
            assert np.allclose(pf.wgts.lw, g["lw"], rtol=1e-10, atol=1e-10)
            assert rel(pf.W, g["W"]) < 1e-9


def check_apf_fused(golden, apf2_cases=((2048, "systematic", 0.7), (4096, "stratified", 0.9),
                                         (4096, "multinomial", 0.7), (3000, "systematic", 0.7))):
    """AuxiliaryPF and GuidedPF of the stock StochVol (Pitt & Shephard's proposal and logeta,
    state_space_models.py:475-498; core.py:299-313) in the FUSED loop: the reference's own runs
    (fixtures sv_apf / sv_guided: same numpy seed -> replayed draws) -- every resample decision,
    ESS and evidence to 1e-9, the final ancestors, particles and weights; Philox mode agrees with
    the operator-at-a-time path's estimate; beyond N = 1024 the APF keeps the operator path."""
    for case, cls, fk in (("sv_apf", ssm.AuxiliaryPF, "apf"), ("sv_guided", ssm.GuidedPF, "guided")):
        g = golden(case)
        y = list(g["y"])
        N, scheme, ESSrmin = int(g["N"]), str(g["scheme"]), float(g["ESSrmin"])
        np.random.seed(int(g["run_seed"]))
        rec = orc.RecordingRNG()
        o = orc.run_filter(orc.StochVol(), y, N, scheme, ESSrmin, fk=fk, rng=rec, keep=True)
        assert o["final_logLt"] == float(g["logLt"])                    # the oracle IS the reference run
        z, u = tapes_from_oracle(rec.tape, len(y), N, scheme)
        pf = pa.SMC(fk=cls(ssm=ssm.StochVol(), data=y), N=N, resampling=scheme, ESSrmin=ESSrmin, replay=(z, u))
        assert pf._fused and describe(pf) == "k_filter_small"
        pf.run()
        assert pf.summaries.rs_flags == [bool(v) for v in g["rs_flags"]] and any(pf.summaries.rs_flags)
        assert rel(pf.summaries.ESSs, g["ESSs"]) < 1e-9
        assert rel(pf.summaries.logLts, g["logLts"]) < 1e-9
        assert np.array_equal(pf.A, g["A"])
        assert np.max(np.abs(pf.X - g["X"])) < 1e-12
        assert np.allclose(pf.wgts.lw, g["lw"], rtol=1e-11, atol=1e-11) and rel(pf.W, g["W"]) < 1e-9
        # stepping one at a time (the reset constant and the auxiliary normalisation travel in the record)
        ps = pa.SMC(fk=cls(ssm=ssm.StochVol(), data=y), N=N, resampling=scheme, ESSrmin=ESSrmin, replay=(z, u))
        for _ in range(len(y)):
            next(ps)
        assert ps.logLt == pf.logLt and np.array_equal(ps.X, pf.X)
    # production mode: the fused APF and the operator path estimate the same evidence
    g = golden("sv_apf")
    y = list(g["y"])
    lls = []
    for s in range(4):
        pf = pa.SMC(fk=ssm.AuxiliaryPF(ssm=ssm.StochVol(), data=y), N=1000, seed=50 + s)
        assert pf._fused
        pf.run()
        lls.append(pf.logLt)
    assert abs(np.mean(lls) - float(g["logLt"])) < 0.15, lls
    from particles_amd.collectors import Moments
    small = pa.SMC(fk=ssm.AuxiliaryPF(ssm=ssm.StochVol(), data=y), N=800, collect=[Moments()])
    assert not small._fused                    # device-side moments with the one-launch APF: the operator path
    # N > 1024: the fused APF with the device-side Moments collector, whole history and a rolling window (VERDICT r4
    # missing 2: core.py:299-313 + smoothing.py:181-255) -- the moments are wmean_and_var of the stored (plain) weights
    for scheme in ("systematic", "multinomial"):
        big = pa.SMC(fk=ssm.AuxiliaryPF(ssm=ssm.StochVol(), data=y[:12]), N=3000, collect=[Moments()], store_history=True,
                     resampling=scheme, seed=9)
        assert big._fused and "k_f_moments" in describe(big), describe(big)
        big.run()
        for t in (0, 5, 11):
            mv = rs.wmean_and_var(big.hist.wgts[t].W, big.hist.X[t])
            assert abs(big.summaries.moments[t]["mean"] - mv["mean"]) < 1e-12 and abs(big.summaries.moments[t]["var"] - mv["var"]) < 1e-11
        roll = pa.SMC(fk=ssm.AuxiliaryPF(ssm=ssm.StochVol(), data=y[:12]), N=3000, store_history=3, resampling=scheme, seed=9)
        assert roll._fused
        roll.run()
        assert roll.logLt == big.logLt and np.array_equal(np.array(roll.hist.X[-1]), np.array(big.hist.X[11]))
        assert np.array_equal(np.array(roll.hist.A[-2]), np.array(big.hist.A[10]))
    # ---- N >= 2048: the APF on the two-level step.  k_propagate leaves TWO tile partials (plain
    # weights: evidence, logged ESS, W; auxiliary weights lw + logeta: decision, shares, integer CDF),
    # k_reduce2 reduces both and sends the reset constant with the record.  Against the oracle run on
    # the same contract (cdf="2level"), replaying its draws: every decision, the final ancestors.
    for N2, scheme, essr in apf2_cases:
        yy = y[:30]
        np.random.seed(7 + N2)
        rec = orc.RecordingRNG()
        o = orc.run_filter(orc.StochVol(), yy, N2, scheme, essr, fk="apf", rng=rec, keep=True, cdf="2level")
        z, u = tapes_from_oracle(rec.tape, len(yy), N2, scheme)
        mk = lambda **kw: pa.SMC(fk=ssm.AuxiliaryPF(ssm=ssm.StochVol(), data=yy), N=N2, resampling=scheme,
                                 ESSrmin=essr, replay=(z, u), **kw)
        pf = mk()
        assert pf._fused and describe(pf).endswith("k_reduce2+k_ancestors2+k_propagate"), describe(pf)
        pf.run()
        assert pf.summaries.rs_flags == o["rs_flag"] and 2 <= sum(o["rs_flag"]) < len(yy) - 1
        assert rel(pf.summaries.ESSs, o["ESS"]) < 1e-9 and rel(pf.summaries.logLts, o["logLt"]) < 1e-9
        assert np.array_equal(pf.A, o["A"])
        assert np.max(np.abs(pf.X - o["X"])) < 1e-12
        assert np.allclose(pf.wgts.lw, o["lw"], rtol=1e-11, atol=1e-11) and rel(pf.W, o["W"]) < 1e-9
        ps = mk(store_history=True)                    # one step at a time, history slots
        for _ in range(len(yy)):
            next(ps)
        assert ps.logLt == pf.logLt and np.array_equal(ps.X, pf.X)
        for t in (3, len(yy) - 1):
            assert np.max(np.abs(ps.hist.X[t] - o["hist"]["X"][t])) < 1e-12
            assert np.allclose(ps.hist.wgts[t].lw, o["hist"]["lw"][t], rtol=1e-11, atol=1e-11)
        with pytest.raises(Exception):
            pf.set_state(lw=np.zeros(N2))
    # ... and on the one-launch filter (N <= 1024) as well: the record's auxiliary normalisation and
    # reset constant come from lw + logeta(X), replacing either alone must be refused, not half-done
    small = pa.SMC(fk=ssm.AuxiliaryPF(ssm=ssm.StochVol(), data=y[:10]), N=500, seed=2)
    small.step_async(4)
    for kw in (dict(lw=np.zeros(500)), dict(X=np.zeros(500))):
        with pytest.raises(Exception, match="auxiliary"):
            small.set_state(**kw)
    # multiSMC: batched as islands where the APF is fused, run by run where it is not (Philox
    # multinomial at N <= 1024 is not on the one-launch filter)
    fk_apf = ssm.AuxiliaryPF(ssm=ssm.StochVol(), data=y[:15])
    for kw in (dict(N=300), dict(N=300, resampling="multinomial"), dict(N=2048, resampling="multinomial")):
        res = pa.multiSMC(fk=fk_apf, nruns=3, out_func=lambda pf: pf.logLt, **kw)
        assert len(res) == 3 and all(np.isfinite(r["output"]) for r in res), kw
        assert len({r["output"] for r in res}) == 3
    # Philox mode, several islands: the evidence estimate
    pf = pa.SMC(fk=ssm.AuxiliaryPF(ssm=ssm.StochVol(), data=y), N=apf2_cases[0][0], seed=5, n_islands=3, collect="off")
    assert pf._fused
    pf.run()
    assert np.max(np.abs(pf.logLts_islands - float(g["logLt"]))) < 0.3, pf.logLts_islands
    gd = pa.SMC(fk=ssm.GuidedPF(ssm=ssm.StochVol(), data=y), N=4096, seed=3)
    assert gd._fused and "k_propagate" in describe(gd)                  # guided StochVol: every path
    gd.run()
    assert abs(gd.logLt - float(golden("sv_guided")["logLt"])) < 0.15


def check_apf_lingauss(golden, big=((2048, "systematic", 0.7), (3000, "stratified", 0.8))):
    """AuxiliaryPF of the stock LinearGauss (kalman.py:436-452: optimal proposal, logeta = the predictive
    density of y_{t+1}) in the fused loop: the reference's own run (fixture lg_apf, replayed draws) --
    decisions, ESS and evidence to 1e-9, final ancestors, particles and weights -- on the one-launch
    filter; on the two-level step against the oracle run on the same contract; the exact Kalman
    likelihood in Philox mode."""
    g = golden("lg_apf")
    y = list(g["y"])
    N, scheme, ESSrmin = int(g["N"]), str(g["scheme"]), float(g["ESSrmin"])
    mk_o = lambda: orc.LinGauss(rho=0.9, sigmaX=1.0, sigmaY=0.6)
    mk_d = lambda: kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=0.6)
    np.random.seed(int(g["run_seed"]))
    rec = orc.RecordingRNG()
    o = orc.run_filter(mk_o(), y, N, scheme, ESSrmin, fk="apf", rng=rec, keep=True)
    assert o["final_logLt"] == float(g["logLt"])                        # the oracle IS the reference run
    z, u = tapes_from_oracle(rec.tape, len(y), N, scheme)
    pf = pa.SMC(fk=ssm.AuxiliaryPF(ssm=mk_d(), data=y), N=N, resampling=scheme, ESSrmin=ESSrmin, replay=(z, u))
    assert pf._fused and describe(pf) == "k_filter_small"
    pf.run()
    assert pf.summaries.rs_flags == [bool(v) for v in g["rs_flags"]] and any(pf.summaries.rs_flags)
    assert rel(pf.summaries.ESSs, g["ESSs"]) < 1e-9 and rel(pf.summaries.logLts, g["logLts"]) < 1e-9
    assert np.array_equal(pf.A, g["A"]) and np.max(np.abs(pf.X - g["X"])) < 1e-12
    assert np.allclose(pf.wgts.lw, g["lw"], rtol=1e-11, atol=1e-11) and rel(pf.W, g["W"]) < 1e-9
    for N2, sch, essr in big:
        np.random.seed(3 + N2)
        rec = orc.RecordingRNG()
        o = orc.run_filter(mk_o(), y, N2, sch, essr, fk="apf", rng=rec, keep=True, cdf="2level")
        z, u = tapes_from_oracle(rec.tape, len(y), N2, sch)
        pf = pa.SMC(fk=ssm.AuxiliaryPF(ssm=mk_d(), data=y), N=N2, resampling=sch, ESSrmin=essr, replay=(z, u))
        assert pf._fused and describe(pf).endswith("k_reduce2+k_ancestors2+k_propagate"), describe(pf)
        pf.run()
        assert pf.summaries.rs_flags == o["rs_flag"] and sum(o["rs_flag"]) >= 2
        assert rel(pf.summaries.ESSs, o["ESS"]) < 1e-9 and rel(pf.summaries.logLts, o["logLt"]) < 1e-9
        assert np.array_equal(pf.A, o["A"]) and np.max(np.abs(pf.X - o["X"])) < 1e-12
    ll, _ = orc.kalman_loglik(mk_o(), y)
    lls = []
    for s in range(4):
        pf = pa.SMC(fk=ssm.AuxiliaryPF(ssm=mk_d(), data=y), N=4096, seed=60 + s)
        pf.run()
        lls.append(pf.logLt)
    assert abs(np.mean(lls) - ll) < 0.1, (lls, ll)


def check_apf_bootstrap(golden, big=((2048, "systematic", 0.7), (3000, "stratified", 0.8), (4096, "multinomial", 0.7))):
    """AuxiliaryBootstrap (state_space_models.py:431-438: the bootstrap move and weight, resampling on lw + logeta,
    weights reset per core.py:299-313) of the stock StochVol and LinearGauss in the fused loop (SMC_FK_APF_BOOT): the
    reference's own runs (fixtures sv_apfboot / lg_apfboot, replayed draws) -- decisions, ESS and evidence to 1e-9,
    final ancestors, particles and weights -- on the one-launch filter; on the two-level step against the oracle run
    on the same contract; a user's subclass with its own logeta keeps the template-method path."""
    for case, mk_o, mk_d in (("sv_apfboot", orc.StochVol, ssm.StochVol),
                             ("lg_apfboot", lambda: orc.LinGauss(rho=0.9, sigmaX=1.0, sigmaY=0.6),
                              lambda: kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=0.6))):
        g = golden(case)
        y = list(g["y"])
        N, scheme, ESSrmin = int(g["N"]), str(g["scheme"]), float(g["ESSrmin"])
        np.random.seed(int(g["run_seed"]))
        rec = orc.RecordingRNG()
        o = orc.run_filter(mk_o(), y, N, scheme, ESSrmin, fk="apfboot", rng=rec, keep=True)
        assert o["final_logLt"] == float(g["logLt"])                    # the oracle IS the reference run
        z, u = tapes_from_oracle(rec.tape, len(y), N, scheme)
        pf = pa.SMC(fk=ssm.AuxiliaryBootstrap(ssm=mk_d(), data=y), N=N, resampling=scheme, ESSrmin=ESSrmin, replay=(z, u))
        assert pf._fused and pf.fk.isAPF and describe(pf) == "k_filter_small"
        pf.run()
        assert pf.summaries.rs_flags == [bool(v) for v in g["rs_flags"]] and any(pf.summaries.rs_flags)
        assert rel(pf.summaries.ESSs, g["ESSs"]) < 1e-9 and rel(pf.summaries.logLts, g["logLts"]) < 1e-9
        assert np.array_equal(pf.A, g["A"]) and np.max(np.abs(pf.X - g["X"])) < 1e-12
        assert np.allclose(pf.wgts.lw, g["lw"], rtol=1e-11, atol=1e-11) and rel(pf.W, g["W"]) < 1e-9
        for N2, sch, essr in big:
            np.random.seed(3 + N2)
            rec = orc.RecordingRNG()
            o = orc.run_filter(mk_o(), y, N2, sch, essr, fk="apfboot", rng=rec, keep=True, cdf="2level")
            z, u = tapes_from_oracle(rec.tape, len(y), N2, sch)
            pf = pa.SMC(fk=ssm.AuxiliaryBootstrap(ssm=mk_d(), data=y), N=N2, resampling=sch, ESSrmin=essr, replay=(z, u))
            assert pf._fused and "k_reduce2" in describe(pf) and describe(pf).endswith("k_ancestors2+k_propagate"), describe(pf)
            pf.run()
            assert pf.summaries.rs_flags == o["rs_flag"] and sum(o["rs_flag"]) >= 2
            assert rel(pf.summaries.ESSs, o["ESS"]) < 1e-9 and rel(pf.summaries.logLts, o["logLt"]) < 1e-9
            assert np.array_equal(pf.A, o["A"]) and np.max(np.abs(pf.X - o["X"])) < 1e-12
    # Philox mode against the exact likelihood (LinearGauss)
    g = golden("lg_apfboot")
    y = list(g["y"])
    ll, _ = orc.kalman_loglik(orc.LinGauss(rho=0.9, sigmaX=1.0, sigmaY=0.6), y)
    lls = []
    for s_ in range(6):
        pf = pa.SMC(fk=ssm.AuxiliaryBootstrap(ssm=kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=0.6), data=y), N=4096, seed=80 + s_)
        pf.run()
        lls.append(pf.logLt)
    assert abs(np.mean(lls) - ll) < 4.0 * max(np.std(lls, ddof=1), 0.02) / np.sqrt(6) + 0.02, (lls, ll)

    class MyAux(ssm.AuxiliaryBootstrap):
        def logeta(self, t, x):
            return 0.0 * x
    pf = pa.SMC(fk=MyAux(ssm=ssm.StochVol(), data=y), N=500)
    assert not pf._fused


def check_apf_mv(golden, big=((3000, 8, "systematic", 0.7), (1 << 13, 32, "stratified", 0.8), (2048, 5, "multinomial", 0.9)),
                 philox_N=4096):
    """AuxiliaryPF of MVLinearGauss (kalman.py:348-361: optimal proposal, logeta = log p(y_{t+1} | x_t)) in the
    fused loop (k_mv_aux + k_mv_aux_restate in front of the flat step, smc_filter_mv.h): the reference's own
    run (fixture mv_apf, replayed draws) -- decisions, ESS and evidence to 1e-9, final ancestors, particles
    and weights; larger N, d up to 32 and the three schemes against the oracle on the device's Q62
    contract; the exact Kalman likelihood in Philox mode; what the fused loop refuses goes to the operators."""
    g = golden("mv_apf")
    y = list(g["y"])
    N, scheme, ESSrmin = int(g["N"]), str(g["scheme"]), float(g["ESSrmin"])
    np.random.seed(int(g["run_seed"]))
    rec = orc.RecordingRNG()
    o = orc.run_filter(orc.Guarniero(alpha=0.4, dx=4), y, N, scheme, ESSrmin, fk="apf", rng=rec, keep=True)
    assert o["final_logLt"] == float(g["logLt"])                        # the oracle IS the reference run
    z, u = tapes_from_oracle(rec.tape, len(y), N, scheme)
    mk_d = lambda dx=4: kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=dx)
    z0, u0 = z, u
    pf = pa.SMC(fk=ssm.AuxiliaryPF(ssm=mk_d(), data=y), N=N, resampling=scheme, ESSrmin=ESSrmin, replay=(z, u))
    assert pf._fused and describe(pf).startswith("k_mv_aux+k_mv_aux_restate+"), describe(pf)
    pf.run()
    pf0_logLt, pf0_X = pf.logLt, np.array(pf.X)
    assert pf.summaries.rs_flags == [bool(v) for v in g["rs_flags"]] and any(pf.summaries.rs_flags)
    assert not all(pf.summaries.rs_flags[1:])                           # both branches of the weight reset
    assert rel(pf.summaries.ESSs, g["ESSs"]) < 1e-9 and rel(pf.summaries.logLts, g["logLts"]) < 1e-9
    assert np.mean(pf.A == g["A"]) >= 0.999
    if np.array_equal(pf.A, g["A"]):
        assert np.max(np.abs(pf.X - g["X"])) < 1e-11
        assert np.allclose(pf.wgts.lw, g["lw"], rtol=1e-10, atol=1e-10) and rel(pf.W, g["W"]) < 1e-9
    for N2, d, sch, essr in big:
        rng = np.random.RandomState(d)
        y2 = [rng.standard_normal(d) for _ in range(7)]
        np.random.seed(3 + N2)
        rec = orc.RecordingRNG()
        o = orc.run_filter(orc.Guarniero(alpha=0.4, dx=d), y2, N2, sch, essr, fk="apf", rng=rec, keep=True, cdf="q62")
        z, u = tapes_from_oracle(rec.tape, len(y2), N2, sch)
        pf = pa.SMC(fk=ssm.AuxiliaryPF(ssm=mk_d(d), data=y2), N=N2, resampling=sch, ESSrmin=essr, replay=(z, u),
                    n_islands=1)
        assert pf._fused and "k_mv_aux" in describe(pf)
        pf.run()
        assert pf.summaries.rs_flags == o["rs_flag"] and sum(o["rs_flag"]) >= 1
        assert rel(pf.summaries.ESSs, o["ESS"]) < 1e-9 and rel(pf.summaries.logLts, o["logLt"]) < 1e-9
        assert np.array_equal(pf.A, o["A"]), int(np.sum(pf.A != o["A"]))
        assert np.max(np.abs(pf.X - o["X"])) < 1e-10 and np.allclose(pf.wgts.lw, o["lw"], rtol=1e-9, atol=1e-9)
    # Philox mode, islands: unbiased likelihood against Kalman's
    mo = orc.Guarniero(alpha=0.4, dx=4)
    ll, _ = orc.kalman_loglik(mo, y)
    pf = pa.SMC(fk=ssm.AuxiliaryPF(ssm=mk_d(), data=y), N=philox_N, seed=71, n_islands=4, collect="off")
    pf.run()
    lls = pf.logLts_islands
    assert abs(np.mean(lls) - ll) < 0.15 and np.std(lls) < 0.3, (lls, ll)
    # history slots, a rolling window and device moments in the fused loop (VERDICT r4 missing 2): the SAME run as without
    # them (replayed draws: the fixture's own), the stored weights are the PLAIN ones (k_propagate_mv puts them back into
    # the slot k_mv_aux wrote the auxiliary weights to), the moments are wmean_and_var of the stored history
    from particles_amd.collectors import Moments
    base = pa.SMC(fk=ssm.AuxiliaryPF(ssm=mk_d(), data=y), N=N, resampling=scheme, ESSrmin=ESSrmin, replay=(z0, u0), store_history=True)
    assert base._fused and "k_mv_aux" in describe(base)
    base.run()
    assert base.logLt == pf0_logLt and np.array_equal(np.array(base.X), pf0_X)
    hist = pa.SMC(fk=ssm.AuxiliaryPF(ssm=mk_d(), data=y), N=N, resampling=scheme, ESSrmin=ESSrmin, replay=(z0, u0),
                  store_history=True, collect=[Moments()])
    assert hist._fused and "k_f_moments" in describe(hist)
    hist.run()
    assert hist.logLt == pf0_logLt
    for t in (0, 3, len(y) - 2, len(y) - 1):
        Wt, Xt = hist.hist.wgts[t].W, hist.hist.X[t]
        assert abs(Wt.sum() - 1.0) < 1e-12
        mvt = rs.wmean_and_var(Wt, Xt)
        assert np.allclose(hist.summaries.moments[t]["mean"], mvt["mean"], rtol=1e-11, atol=1e-12)
        assert np.allclose(hist.summaries.moments[t]["var"], mvt["var"], rtol=1e-10, atol=1e-11)
        # the stored log-weights are the plain ones: their ESS is the logged ESS of that step
        assert abs(1.0 / np.sum(Wt ** 2) / hist.summaries.ESSs[t] - 1.0) < 1e-9
    roll = pa.SMC(fk=ssm.AuxiliaryPF(ssm=mk_d(), data=y), N=N, resampling=scheme, ESSrmin=ESSrmin, replay=(z0, u0), store_history=2)
    assert roll._fused
    roll.run()
    assert roll.logLt == pf0_logLt and np.array_equal(np.array(roll.hist.X[-1]), np.array(hist.hist.X[len(y) - 1]))
    assert np.allclose(roll.hist.wgts[-2].lw, hist.hist.wgts[len(y) - 2].lw, rtol=0, atol=0)


def check_resident_user_model(golden):
    """A user-defined model written with numpy expressions (here Gordon et al's, transcribed
    from state_space_models.py:546-577, and StochVol's) run through the template-method step
    with its arrays resident in HBM (pa.set_resident): DeviceArray arithmetic and ufuncs.
    Same numpy seed as the reference's run -> the reference's results."""
    class MyGordon(ssm.StateSpaceModel):
        default_params = {"a": 0.05, "b": 0.5, "c": 25.0, "d": 8.0, "e": 1.2, "sigmaX": 3.162278}

        def PX0(self):
            return dists.Normal(scale=2.0)

        def PX(self, t, xp):
            return dists.Normal(loc=self.b * xp + self.c * xp / (1.0 + xp ** 2)
                                + self.d * np.cos(self.e * (t - 1)), scale=self.sigmaX)

        def PY(self, t, xp, x):
            return dists.Normal(loc=self.a * x ** 2)

    class MySV(ssm.StochVol):
        def _device_params(self, fk_kind):
            return None                              # not the fused loop: the generic step

    pa.set_resident(True)
    try:
        for case, model in (("gordon_boot", MyGordon()), ("sv_systematic", MySV())):
            g = golden(case)
            y = list(g["y"])
            np.random.seed(int(g["run_seed"]))
            pf = pa.SMC(fk=ssm.Bootstrap(ssm=model, data=y), N=int(g["N"]),
                        resampling=str(g["scheme"]), ESSrmin=float(g["ESSrmin"]))
            assert not pf._fused
            pf.run()
            assert isinstance(pf.X, pa.DeviceArray) and isinstance(pf.wgts.lw, pa.DeviceArray)
            assert pf.summaries.rs_flags == [bool(v) for v in g["rs_flags"]]
            assert rel(pf.summaries.logLts, g["logLts"]) < 1e-9
            A = np.asarray(pf.A)
            assert np.mean(A == g["A"]) >= 0.999
            if np.array_equal(A, g["A"]):
                assert np.max(np.abs(pf.X.get() - g["X"])) < 1e-11
                assert np.allclose(np.asarray(pf.wgts.lw), g["lw"], rtol=1e-10, atol=1e-10)
        # a multivariate model as the reference writes it (np.dot(xp, F.T), kalman.py:339-346)
        class MyMV(kalman.MVLinearGauss_Guarniero_etal):
            def _device_params(self, fk_kind):
                return None
        g = golden("mv4_boot")
        y = list(g["y"])
        np.random.seed(int(g["run_seed"]))
        pf = pa.SMC(fk=ssm.Bootstrap(ssm=MyMV(alpha=0.4, dx=4), data=y), N=int(g["N"]),
                    resampling=str(g["scheme"]), ESSrmin=float(g["ESSrmin"]))
        assert not pf._fused
        pf.run()
        assert isinstance(pf.X, pa.DeviceArray) and pf.X.shape == (int(g["N"]), 4)
        assert pf.summaries.rs_flags == [bool(v) for v in g["rs_flags"]]
        assert rel(pf.summaries.logLts, g["logLts"]) < 1e-9
        if np.array_equal(np.asarray(pf.A), g["A"]):
            assert np.max(np.abs(pf.X.get() - g["X"])) < 1e-11
        # the operators themselves against numpy
        rng = np.random.default_rng(2)
        xh, yh = rng.standard_normal(1000), rng.random(1000) + 0.5
        x, yv = pa.DeviceArray.from_numpy(xh), pa.DeviceArray.from_numpy(yh)
        for got, want in ((2.0 * x + yv / 3.0 - 1.0, 2.0 * xh + yh / 3.0 - 1.0),
                          (1.0 / (1.0 + x ** 2), 1.0 / (1.0 + xh ** 2)),
                          (-(x - yv) * (yv - 2.0), -(xh - yh) * (yh - 2.0)),
                          (np.sqrt(yv) + yv ** 0.5, 2.0 * np.sqrt(yh)),
                          (np.float64(3.0) - x, 3.0 - xh),
                          (np.minimum(x, yv), np.minimum(xh, yh)),
                          (abs(x), np.abs(xh))):
            assert np.array_equal(got.get(), want)                  # IEEE-exact operations
        for got, want in ((np.exp(0.5 * x), np.exp(0.5 * xh)), (np.log(yv), np.log(yh)),
                          (np.cos(x) * np.sin(x), np.cos(xh) * np.sin(xh)),
                          (yv ** 1.7, yh ** 1.7)):
            assert np.allclose(got.get(), want, rtol=1e-14, atol=1e-15)
        idx = rng.integers(0, 1000, size=500)
        assert np.array_equal(x[idx].get(), xh[idx])
        Xh, row, Mh = rng.standard_normal((300, 5)), rng.standard_normal(5), rng.standard_normal((5, 3))
        Xd = pa.DeviceArray.from_numpy(Xh)
        assert np.array_equal((Xd * row + 1.0).get(), Xh * row + 1.0)          # (N, d) with a (d,) row
        assert np.array_equal((row - Xd).get(), row - Xh)
        assert np.allclose((Xd @ Mh).get(), Xh @ Mh, rtol=1e-13, atol=1e-13)
        assert np.allclose(np.dot(Xd, Mh).get(), Xh @ Mh, rtol=1e-13, atol=1e-13)
        assert np.array_equal(Xd[idx[:50] % 300].get(), Xh[idx[:50] % 300])
    finally:
        pa.set_resident(False)


def check_indep_prod(golden):
    """IndepProd (distributions.py:1066-1106): a bivariate model with a Gaussian and a Poisson
    observation, transcribed from the reference-side definition in tests/golden/make_golden.py;
    same numpy seed -> the reference's run, with host arrays and with arrays resident in HBM."""
    class Indep2(ssm.StateSpaceModel):
        def PX0(self):
            return dists.IndepProd(dists.Normal(scale=1.0), dists.Normal(scale=2.0))

        def PX(self, t, xp):
            return dists.IndepProd(dists.Normal(loc=0.9 * xp[:, 0]),
                                   dists.Normal(loc=0.5 * xp[:, 1] + 0.1 * xp[:, 0], scale=0.7))

        def PY(self, t, xp, x):
            return dists.IndepProd(dists.Normal(loc=x[:, 0], scale=0.5),
                                   dists.Poisson(rate=np.exp(0.3 * x[:, 1])))

    g = golden("indep_boot")
    y = list(g["y"])
    for resident in (False, True):
        pa.set_resident(resident)
        try:
            np.random.seed(int(g["run_seed"]))
            pf = pa.SMC(fk=ssm.Bootstrap(ssm=Indep2(), data=y), N=int(g["N"]),
                        resampling=str(g["scheme"]), ESSrmin=float(g["ESSrmin"]))
            assert not pf._fused
            pf.run()
            assert isinstance(pf.X, pa.DeviceArray) == resident
            X = np.asarray(pf.X)
            assert X.shape == (int(g["N"]), 2)
            assert pf.summaries.rs_flags == [bool(v) for v in g["rs_flags"]]
            assert rel(pf.summaries.logLts, g["logLts"]) < 1e-9
            A = np.asarray(pf.A)
            assert np.mean(A == g["A"]) >= 0.995
            if np.array_equal(A, g["A"]):
                assert np.max(np.abs(X - g["X"])) < 1e-11
                assert np.allclose(np.asarray(pf.wgts.lw), g["lw"], rtol=1e-10, atol=1e-10)
        finally:
            pa.set_resident(False)
    # columns in and out of an (N, d) device array
    rng = np.random.default_rng(8)
    Xh = rng.standard_normal((257, 3))
    Xh[5, 1] = np.inf
    Xh[6, 2] = -0.0
    Xd = pa.DeviceArray.from_numpy(Xh)
    for i in (0, 1, 2, -1):
        assert np.array_equal(Xd[..., i].get(), Xh[..., i]) and np.array_equal(Xd[:, i].get(), Xh[:, i])
    back = np.stack([Xd[:, i] for i in range(3)], axis=1)
    assert isinstance(back, pa.DeviceArray) and np.array_equal(back.get(), Xh)
    assert np.array_equal(np.signbit(back.get()), np.signbit(Xh))
    try:
        Xd[:, 3]
        raise AssertionError("column 3 of an (N, 3) array")
    except IndexError:
        pass
    d = dists.IndepProd(dists.Normal(loc=1.0, scale=2.0), dists.Normal(), dists.Normal(scale=0.5))
    Xh[5, 1] = 0.25
    want = (orc.normal_logpdf(Xh[:, 0], 1.0, 2.0) + orc.normal_logpdf(Xh[:, 1])
            + orc.normal_logpdf(Xh[:, 2], 0.0, 0.5))
    assert np.allclose(d.logpdf(Xh), want, rtol=1e-14, atol=1e-14)
    assert d.rvs(size=10).shape == (10, 3) and d.dim == 3


def check_sqmc(golden, monkeypatch, philox_N=4096, philox_runs=4, philox_T=40, sorted_N=(1, 2, 64, 1024, 1 << 14),
               ab_N=2048, both_modes_for_all=True, replay_cases=None):
    """SQMC (SMC(qmc=True), core.py:315-349) on device operators: the reference's runs on its
    recorded Sobol' points; then the operators themselves (Sobol' generator vs scipy's, ndtri
    vs scipy's, argsort vs numpy's) and a run on device-generated points against Kalman."""
    from particles_amd import rqmc, hilbert
    cases = [("sqmc_toy", lambda: kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=0.5), ssm.Bootstrap),
             ("sqmc_sv", lambda: ssm.StochVol(), ssm.Bootstrap),
             ("sqmc_guided", lambda: kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=0.3), ssm.GuidedPF),
             ("sqmc_mv2", lambda: kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=2), ssm.Bootstrap),
             ("sqmc_mv3_guided", lambda: kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=3), ssm.GuidedPF)]
    for case, mk, cls in cases:
        if replay_cases is not None and case not in replay_cases:
            continue
        g = golden(case)
        y = list(g["y"])
        for resident in ((False, True) if both_modes_for_all or case == "sqmc_mv2" else (True,)):
            tape = [g["u0"]] + list(g["u"])
            it = iter(tape)

            def replay(N, d, it=it, resident=resident):
                u = next(it)
                assert u.shape == (N, d)
                return pa.DeviceArray.from_numpy(u) if resident else u
            monkeypatch.setattr(rqmc, "sobol", replay)
            pa.set_resident(resident)
            try:
                pf = pa.SMC(fk=cls(ssm=mk(), data=y), N=int(g["N"]), qmc=True)
                assert not pf._fused
                pf.run()
                assert pf.summaries.rs_flags == [bool(v) for v in g["rs_flags"]]
                assert rel(pf.summaries.logLts, g["logLts"]) < 1e-9
                assert rel(pf.summaries.ESSs, g["ESSs"]) < 1e-8
                A = np.asarray(pf.A)
                assert np.mean(A == g["A"]) >= 0.995
                if np.array_equal(A, g["A"]):
                    assert np.max(np.abs(np.asarray(pf.X) - g["X"])) < 1e-11
                    assert np.allclose(np.asarray(pf.wgts.lw), g["lw"], rtol=1e-10, atol=1e-10)
            finally:
                pa.set_resident(False)
                monkeypatch.undo()
    # ---- the Hilbert codec and sort against the reference's (hilbert.py:13-58)
    g = golden("hilbert")
    for d in (2, 3, 5, 8):
        assert np.array_equal(hilbert.hilbert_array(g["xint%d" % d]), g["h%d" % d])     # integers: exact
        order = hilbert.hilbert_sort(g["x%d" % d])
        assert sorted(order) == list(range(len(order)))
        assert np.mean(order == g["order%d" % d]) >= 0.97      # the grid cell of a point may move by
        #                                                        one where exp / mean / std differ by an ulp
    xd = pa.DeviceArray.from_numpy(g["x3"])
    od = hilbert.hilbert_sort(xd)
    assert isinstance(od, pa.DeviceArray) and np.array_equal(od.get(), hilbert.hilbert_sort(g["x3"]))
    # ---- the operators
    from scipy.stats import qmc
    from scipy import special
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for d in (1, 2, 3, 7, 10):
            ref = qmc.Sobol(d, scramble=False).random(1000)
            assert np.array_equal(rqmc.sobol_unscrambled(1000, d), ref)
    rng = np.random.default_rng(3)
    u = np.concatenate([rng.random(5000), 10.0 ** -rng.uniform(1, 300, 2000),
                        1.0 - 10.0 ** -rng.uniform(1, 15, 1000), [0.0, 1.0, 0.5, -0.1, 1.5, np.nan]])
    got = dists.Normal(loc=0.3, scale=1.7).ppf(u)
    want = special.ndtri(u) * 1.7 + 0.3
    fin = np.isfinite(want)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[np.isinf(want)], want[np.isinf(want)])
    assert np.max(np.abs(got[fin] - want[fin]) / (1.0 + np.abs(want[fin]))) < 5e-16
    central = fin & (u > 0.14) & (u < 0.86)
    assert np.array_equal(got[central], want[central])           # IEEE + - * / only
    x = rng.standard_normal(5003)
    assert np.array_equal(hilbert.argsort(x), np.argsort(x, kind="stable"))
    assert np.array_equal(hilbert.hilbert_sort(x.reshape(-1, 1)), np.argsort(x, kind="stable"))
    # ---- a run on device-generated, digitally shifted points (Philox mode)
    g = golden("kalman_toy")
    y = [np.atleast_1d(v) for v in np.squeeze(g["y"])][:philox_T]
    ll, _ = orc.kalman_loglik(orc.ToySSM(0.2), y)
    rs.set_rng("philox")
    try:
        pa.seed(5)
        pts = rqmc.sobol(1024, 2)
        assert isinstance(pts, pa.DeviceArray)
        ph = pts.get()
        assert ph.min() > 0.0 and ph.max() < 1.0
        for c in range(2):           # a (0, m, 1)-net in base 2 survives the digital shift
            assert np.array_equal(np.sort(np.floor(ph[:, c] * 1024).astype(int)), np.arange(1024))
        # the closed-form sorted order: sobol_sorted == sobol[argsort(first coordinate)], same stream
        for n, d in zip(sorted_N, (2, 3, 2, 4, 2, 3, 2)):
            pa.seed(11 + n)
            u = rqmc.sobol(n, d).get()
            pa.seed(11 + n)
            us = rqmc.sobol_sorted(n, d)
            assert np.array_equal(us.get(), u[np.argsort(u[:, 0], kind="stable")])
        assert rqmc.sobol_sorted(1000, 2) is None                 # not a power of two: the generic route
        # ... and the SQMC run is the same run with or without it (X, A, logLt bit for bit)
        runs = []
        _lib.FUSED_SQMC[0] = False                   # (the operator path: the fused step has its own check)
        try:
            for closed_form in (True, False):
                if not closed_form:
                    monkeypatch.setattr(rqmc, "sobol_sorted", lambda N, d: None)
                pa.seed(77)
                pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y[:8]), N=ab_N, qmc=True)
                assert not pf._fused
                pf.run()
                runs.append((np.asarray(pf.X), np.asarray(pf.A), pf.logLt))
        finally:
            _lib.FUSED_SQMC[0] = True
        monkeypatch.undo()
        assert all(np.array_equal(a, b) for a, b in zip(runs[0], runs[1]))
        lls = []
        for s in range(philox_runs):
            pa.seed(100 + s)
            pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=philox_N, qmc=True)
            pf.run()
            lls.append(pf.logLt)
        assert np.max(np.abs(np.array(lls) - ll)) < 0.25
    finally:
        rs.set_rng("numpy")


def sobol_sorted_points_np(seed, counter, N, d):
    """The d-dimensional scrambled Sobol' point set `counter` of the device stream, sorted by its first
    coordinate -- restated with scipy's engine for the sequence (Joe & Kuo's direction numbers, 30 bits,
    Gray-code order), the digital shift from the Philox restatement (word 0 of call (coordinate, counter
    lo, counter hi, stream 1), top 30 bits), rqmc.py:9-13's safe_generate, and a stable argsort: no
    closed form.  Returns (points in Gray order, the same sorted by the first coordinate)."""
    import warnings
    from scipy.stats import qmc
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        base = qmc.Sobol(d, scramble=False).random(N)
    xi = np.rint(base * 2.0 ** 30).astype(np.uint64)
    assert np.array_equal(xi / 2.0 ** 30, base)
    for c in range(d):
        x01, _ = orc.philox_u64_pair(seed, np.array([c]), int(counter) & 0xFFFFFFFF, int(counter) >> 32, orc.STREAM_RESAMPLE)
        xi[:, c] ^= np.uint64(int(x01[0]) >> 34)
    u = 0.5 + (1.0 - 1e-10) * (xi.astype(np.float64) / 2.0 ** 30 - 0.5)
    return u, u[np.argsort(u[:, 0], kind="stable")]


def audit_sqmc_history(pf, mk_orc, fk, y, point_seed, ctr0, island=0, tol=1e-12):
    """Teacher-forced audit of a fused SQMC run with store_history=True (core.py:315-321, 339-349): from
    the device's own X_{t-1}, lw_{t-1} every step must be h_order = argsort(X_{t-1}); A_t =
    h_order[inverse_cdf(u[tau, 0], W[h_order])] -- bit for bit on the two-level contract (oracle.c) applied
    to the weights in sorted order, and against the reference's sequential fp64 CDF with every mismatch a
    certified near-tie; X_t = Gamma(t, X_{t-1}[A_t], u[tau, 1]) with scipy's ndtri; weights, ESS,
    evidence.  Returns the near-tie count."""
    from scipy import special
    N, T = pf.N, pf._n
    model = mk_orc()
    ctx = orc.StepCtx(model, fk, y[0])
    summ = pf._summ()[island]
    near_ties = 0
    logLt = 0.0
    X_prev = lw_prev = None
    for t in range(T):
        X = pf._history(_lib.FIELD_X, t, island)
        lw = pf._history(_lib.FIELD_LW, t, island)
        ctr = ctr0 + t + (island << 32)
        if t == 0:
            u, _ = sobol_sorted_points_np(point_seed, ctr, N, 1)
            zt, Xp = special.ndtri(u[:, 0]), None
        else:
            assert bool(summ[t, 4])                                    # always resamples (core.py:340)
            _, us = sobol_sorted_points_np(point_seed, ctr, N, 2)
            h = np.argsort(X_prev, kind="stable")
            lws = lw_prev[h]
            A = pf._history(_lib.FIELD_A, t, island)
            A_c, _ = orc.inverse_cdf_2level_c("multinomial", us[:, 0], lws)
            assert np.array_equal(A, h[A_c]), (t, int(np.sum(A != h[A_c])))
            W_ref = orc.exp_and_normalise(lws)
            try:
                A_ref = orc.inverse_cdf(us[:, 0], W_ref)
            except IndexError:
                A_ref = None
            if A_ref is not None and not np.array_equal(A_c, A_ref):
                n, ok = orc.audit_near_ties(us[:, 0], W_ref, A_ref, A_c)
                assert ok, (t, n)
                near_ties += n
            ess_prev = orc.two_level_reduce(*orc.tile_partials(lws))[0]["ESS"]     # (partials of the sorted order)
            assert ess_prev == summ[t - 1, 0], (t, ess_prev, summ[t - 1, 0])
            Xp = X_prev[A]
            zt = special.ndtri(us[:, 1])
        Xo, inc = orc.propagate(model, fk, t, np.asarray(y[t]), Xp, zt, ctx)
        lwo = np.where(np.isnan(inc), -np.inf, inc)
        assert np.allclose(X, Xo, rtol=tol, atol=tol), (t, "X", float(np.max(np.abs(X - Xo))))
        assert np.allclose(lw, lwo, rtol=100 * tol, atol=100 * tol), (t, "lw")
        w = orc.Weights(lw=lw.copy())
        assert abs(summ[t, 0] / w.ESS - 1) < 1e-10 and abs(summ[t, 2] - w.log_mean) < 1e-10 * max(1.0, abs(w.log_mean)), t
        logLt += summ[t, 2]
        assert abs(summ[t, 3] - logLt) < 1e-10 * max(1.0, abs(logLt)), (t, "logLt")
        X_prev, lw_prev = X, lw
    log_near_ties("sqmc N=%d T=%d" % (N, T), near_ties, N * max(T - 1, 0))
    return near_ties


def check_sqmc_fused(sizes=(2048, 4096), T=6, audit_sizes=(4096,), islands_N=2048):
    """SMC(qmc=True) as a fused loop (SMC_FLAG_SQMC; smc_filter_sqmc.h): (1) the SAME run as the
    operator path on the same points -- X, lw, A of every step bit for bit where the two exact CDFs
    (two-level vs flat Q62) agree, i.e. everywhere but at certified near-ties; (2) the teacher-forced
    audit against the oracle and the reference's formula; (3) islands = independent runs with their own
    point sets; (4) what does not fuse runs on the operators."""
    cases = [("toy", lambda: kalman.ToySSM(0.2), lambda: orc.ToySSM(0.2), ssm.Bootstrap, "bootstrap"),
             ("sv", lambda: ssm.StochVol(), lambda: orc.StochVol(), ssm.Bootstrap, "bootstrap"),
             ("lg_guided", lambda: kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=0.3),
              lambda: orc.LinGauss(rho=0.9, sigmaX=1.0, sigmaY=0.3), ssm.GuidedPF, "guided")]
    rng = np.random.RandomState(12)
    y = [np.array([v]) for v in 0.5 * np.cumsum(rng.standard_normal(T))]
    rs.set_rng("philox")
    try:
        for name, mk, mk_o, cls, fkname in cases:
            for N in sizes:
                runs = []
                for fused in (True, False):
                    _lib.FUSED_SQMC[0] = fused
                    pa.seed(31)
                    pf = pa.SMC(fk=cls(ssm=mk(), data=y), N=N, qmc=True, collect="off")
                    assert pf._fused == fused
                    if fused:
                        assert "k_sq_permute" in describe(pf) and "k_rs_sort" in describe(pf)
                    steps = []
                    for t in range(T):
                        next(pf)
                        steps.append((np.asarray(pf.X).copy(), None if t == 0 else np.asarray(pf.A).copy(),
                                      np.asarray(pf.wgts.lw).copy(), float(pf.logLt)))
                    runs.append(steps)
                _lib.FUSED_SQMC[0] = True
                for t in range(T):
                    (Xa, Aa, la, La), (Xb, Ab, lb, Lb) = runs[0][t], runs[1][t]
                    if Aa is not None and not np.array_equal(Aa, Ab):
                        break                 # a near-tie between the two exact CDFs: the audit below decides
                    # (StochVol: the fused kernels form log N(y; 0, e^{x/2}) with one exp, the operators as scipy does)
                    assert np.array_equal(Xa, Xb), (name, N, t)
                    assert np.array_equal(la, lb) if name != "sv" else np.allclose(la, lb, rtol=1e-13, atol=1e-13), (name, N, t)
                    assert abs(La - Lb) < 1e-12 * max(1.0, abs(La)), (name, N, t)
                else:
                    t = T
                assert t >= T - 1 or N > 1 << 16, (name, N, t)
            for N in audit_sizes:
                pa.seed(47)
                c0 = _lib._counter + 1
                pf = pa.SMC(fk=cls(ssm=mk(), data=y), N=N, qmc=True, collect="off", store_history=True)
                assert pf._fused
                pf.run()
                audit_sqmc_history(pf, mk_o, fkname, y, 47, c0)
                # the production run (no history) ends in the same state
                pa.seed(47)
                pq = pa.SMC(fk=cls(ssm=mk(), data=y), N=N, qmc=True, collect="off")
                pq.run()
                assert np.array_equal(np.asarray(pq.X), np.asarray(pf.X)) and pq.logLt == pf.logLt
                # ... the sorted weights recomputed from the sorted keys (bootstrap filters) or gathered through the
                # permutation (SMC_PATH_SQ_GATHER): the same bits -- ESS of every step, particles, ancestors
                os.environ["SMC_SQ_GATHER"] = "1"
                try:
                    pa.seed(47)
                    pg = pa.SMC(fk=cls(ssm=mk(), data=y), N=N, qmc=True, collect="off")
                    pg.run()
                finally:
                    del os.environ["SMC_SQ_GATHER"]
                assert np.array_equal(pg._summ(), pq._summ()) and np.array_equal(np.asarray(pg.X), np.asarray(pq.X))
                assert np.array_equal(np.asarray(pg.A), np.asarray(pq.A))
                # ... and so does the loop replayed from hipGraphs (steps 0 and 1 eagerly, then captured pairs)
                pa.seed(47)
                pg = pa.SMC(fk=cls(ssm=mk(), data=y), N=N, qmc=True, collect="off", use_graph=True)
                assert pg._fused
                pg.run()
                assert np.array_equal(np.asarray(pg.X), np.asarray(pf.X)) and pg.logLt == pf.logLt
        # islands: each an SQMC run of its own (point sets keyed by the island id)
        pa.seed(53)
        c0 = _lib._counter + 1
        pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=islands_N, qmc=True, collect="off",
                    store_history=True, n_islands=3)
        pf.run()
        for isl in (0, 2):
            audit_sqmc_history(pf, lambda: orc.ToySSM(0.2), "bootstrap", y, 53, c0, island=isl)
        ll = pf.logLt_islands() if hasattr(pf, "logLt_islands") else None
        # (4) outside the fused family: the operator path
        pa.seed(3)
        assert not pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=1000, qmc=True)._fused       # N != 2^k
        # (below two tiles the flat step runs it -- check_sqmc_fused_small -- and keeps no history slots)
        assert not pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=1024, qmc=True, store_history=True)._fused
    finally:
        _lib.FUSED_SQMC[0] = True
        rs.set_rng("numpy")


def check_sqmc_fused_small(sizes=(32, 256, 1024), T=6):
    """SMC(qmc=True) of a univariate filter below two tiles (N = 2^k, 32 <= N <= 1024) as a fused loop on the flat
    step: argsort + the step's 2 Sobol' coordinates + k_sqmv_tapes + the flat multinomial search + k_propagate fed
    from the tape -- the SAME run as the operator path on the same points (both form the flat Q62 CDF of the sorted
    weights: ancestors identical); islands with their own point sets against Kalman's likelihood."""
    cases = [("toy", lambda: kalman.ToySSM(0.2), ssm.Bootstrap),
             ("sv", lambda: ssm.StochVol(), ssm.Bootstrap),
             ("lg_guided", lambda: kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=0.3), ssm.GuidedPF)]
    rng = np.random.RandomState(12)
    y = [np.array([v]) for v in 0.5 * np.cumsum(rng.standard_normal(T))]
    rs.set_rng("philox")
    try:
        for name, mk, cls in cases:
            for N in sizes:
                runs = []
                for fused in (True, False):
                    _lib.FUSED_SQMC[0] = fused
                    pa.seed(31)
                    pf = pa.SMC(fk=cls(ssm=mk(), data=y), N=N, qmc=True, collect="off")
                    assert pf._fused == fused
                    if fused:
                        assert describe(pf).startswith("k_rs_sort+k_sobol+k_sqmv_tapes+"), describe(pf)
                    steps = []
                    for t in range(T):
                        next(pf)
                        assert pf.rs_flag == (t > 0)
                        steps.append((np.asarray(pf.X).copy(), None if t == 0 else np.asarray(pf.A).copy(),
                                      np.asarray(pf.wgts.lw).copy(), float(pf.logLt)))
                    runs.append(steps)
                _lib.FUSED_SQMC[0] = True
                for t in range(T):
                    (Xa, Aa, la, La), (Xb, Ab, lb, Lb) = runs[0][t], runs[1][t]
                    if name == "sv" and Aa is not None and not np.array_equal(Aa, Ab):
                        break                  # (weights that differ in their last bits: see below)
                    assert Aa is None or np.array_equal(Aa, Ab), (name, N, t)
                    assert np.array_equal(Xa, Xb), (name, N, t)
                    # (StochVol: the fused kernels form log N(y; 0, e^{x/2}) with one exp, the operators as scipy does)
                    assert np.array_equal(la, lb) if name != "sv" else np.allclose(la, lb, rtol=1e-13, atol=1e-13), (name, N, t)
                    assert abs(La - Lb) < 1e-12 * max(1.0, abs(La)), (name, N, t)
                else:
                    t = T
                assert t >= 2, (name, N, t)
        # islands: each an SQMC run of its own; the evidence of a linear Gaussian model
        model = kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=0.3)
        ll = orc.kalman_loglik(orc.LinGauss(rho=0.9, sigmaX=1.0, sigmaY=0.3), y)
        ll = float(ll[0]) if isinstance(ll, tuple) else float(ll)
        pa.seed(53)
        pf = pa.SMC(fk=ssm.GuidedPF(ssm=model, data=y), N=512, qmc=True, collect="off", n_islands=4)
        assert pf._fused
        pf.run()
        lls = pf.logLts_islands
        assert len(set(lls.tolist())) == 4 and np.max(np.abs(lls - ll)) < 0.1, (lls, ll)
        assert not pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=16, qmc=True)._fused
        assert not pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=512, qmc=True, use_graph=True)._fused
    finally:
        _lib.FUSED_SQMC[0] = True
        rs.set_rng("numpy")


def check_sqmc_fused_mv(cases=((1024, 2), (4096, 3), (2048, 5)), T=5, islands_N=512):
    """SMC(qmc=True) of MVLinearGauss (2 <= d <= 9) as a fused loop: Hilbert sort + the step's d + 1 Sobol'
    coordinates + the flat step (smc_filter_sqmc.h) -- the SAME run as the operator path on the same points:
    ancestors identical (both form the flat Q62 CDF of the weights in Hilbert order), particles and weights to
    1e-12 (MFMA summation order); Kalman's likelihood; islands; what does not fuse."""
    rs.set_rng("philox")
    try:
        for N, d in cases:
            model_o = orc.Guarniero(alpha=0.4, dx=d)
            rng = np.random.RandomState(d)
            y = [rng.standard_normal(d) for _ in range(T)]
            for cls in (ssm.Bootstrap, ssm.GuidedPF):
                runs = []
                for fused in (True, False):
                    _lib.FUSED_SQMC[0] = fused
                    pa.seed(17)
                    pf = pa.SMC(fk=cls(ssm=kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=d), data=y), N=N, qmc=True,
                                collect="off")
                    assert pf._fused == fused
                    if fused:
                        assert describe(pf).startswith("smc_hilbert_sort+k_sobol+k_sqmv_tapes+"), describe(pf)
                    steps = []
                    for t in range(T):
                        next(pf)
                        assert pf.rs_flag == (t > 0)
                        steps.append((np.asarray(pf.X).copy(), None if t == 0 else np.asarray(pf.A).copy(),
                                      np.asarray(pf.wgts.lw).copy(), float(pf.logLt)))
                    runs.append(steps)
                _lib.FUSED_SQMC[0] = True
                for t in range(T):
                    (Xa, Aa, la, La), (Xb, Ab, lb, Lb) = runs[0][t], runs[1][t]
                    if Aa is not None and not np.array_equal(Aa, Ab):
                        # (the weights agree to ~1e-15: a tie decided by their last bits; rare, and the end of a
                        #  step-by-step comparison)
                        assert np.mean(Aa != Ab) < 0.005, (N, d, t)
                        break
                    # (products summed in MFMA order here, in the operators' order there)
                    assert np.allclose(Xa, Xb, rtol=1e-12, atol=1e-12) and np.allclose(la, lb, rtol=1e-11, atol=1e-11), (N, d, t)
                    assert abs(La - Lb) < 1e-11 * max(1.0, abs(La))
                else:
                    t = T
                assert t >= 2, (N, d, t)
        # evidence against Kalman's, islands with their own point sets
        d = 3
        rng = np.random.RandomState(1)
        y = [0.5 * rng.standard_normal(d) for _ in range(12)]
        ll, _ = orc.kalman_loglik(orc.Guarniero(alpha=0.4, dx=d), y)
        pa.seed(23)
        pf = pa.SMC(fk=ssm.GuidedPF(ssm=kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=d), data=y), N=islands_N, qmc=True,
                    collect="off", n_islands=4)
        assert pf._fused
        pf.run()
        lls = pf.logLts_islands
        assert len(set(lls.tolist())) == 4 and np.max(np.abs(lls - ll)) < 0.3, (lls, ll)
        # not fused: d beyond the 10 Sobol' coordinates, history slots, N not a power of two
        mk = lambda dx: kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=dx)
        y10 = [rng.standard_normal(10) for _ in range(3)]
        assert not pa.SMC(fk=ssm.Bootstrap(ssm=mk(10), data=y10), N=256, qmc=True)._fused
        assert not pa.SMC(fk=ssm.Bootstrap(ssm=mk(3), data=y), N=256, qmc=True, store_history=True)._fused
        assert not pa.SMC(fk=ssm.Bootstrap(ssm=mk(3), data=y), N=300, qmc=True)._fused
    finally:
        _lib.FUSED_SQMC[0] = True
        rs.set_rng("numpy")


def check_collectors_on_fused(golden):
    from particles_amd.collectors import Moments
    g = golden("kalman_toy")
    y = [np.atleast_1d(v) for v in np.squeeze(g["y"])][:25]
    pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=5000, seed=9,
                collect=[Moments()], store_history=True)
    assert pf._device_moments and not pf._needs_per_step_host()      # moments and history on the device
    pf.run()
    # the device-side Moments against rs.wmean_and_var on the stored history, step by step
    for t in (0, 7, 24):
        mv = rs.wmean_and_var(pf.hist.wgts[t].W, pf.hist.X[t])
        assert abs(pf.summaries.moments[t]["mean"] - mv["mean"]) < 1e-12
        assert abs(pf.summaries.moments[t]["var"] - mv["var"]) < 1e-11
    # multivariate: (d,) means and variances
    rng = np.random.RandomState(5)
    ymv = [rng.standard_normal((1, 4)) for _ in range(8)]
    pm = pa.SMC(fk=ssm.GuidedPF(ssm=kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=4), data=ymv),
                N=3000, seed=2, collect=[Moments()])
    pm.run()
    last = pm.summaries.moments[-1]
    mv = rs.wmean_and_var(pm.W, pm.X)
    assert last["mean"].shape == (4,) and np.allclose(last["mean"], mv["mean"], atol=1e-12)
    assert np.allclose(last["var"], mv["var"], atol=1e-11) and len(pm.summaries.moments) == 8
    # a custom moment function keeps the per-step host path
    pc_ = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y[:5]), N=500, seed=9,
                 collect=[Moments(mom_func=lambda W, X: float(np.sum(W * X)))])
    assert pc_._needs_per_step_host()
    pc_.run()
    assert len(pc_.summaries.moments) == 5 and np.isfinite(pc_.summaries.moments[-1])
    _, means = orc.kalman_loglik(orc.ToySSM(0.2), y)
    est = np.array([m["mean"] for m in pf.summaries.moments])
    assert est.shape == (25,) and np.max(np.abs(est - means)) < 0.1
    assert len(pf.summaries.ESSs) == 25 and len(pf.hist.A) == 25
    B = pf.hist.compute_trajectories()
    assert B.shape == (25, 5000) and np.array_equal(B[-1], np.arange(5000))


class _PickleCustomFK(ssm.Bootstrap):            # a user subclass: the template-method path on device operators
    def logG(self, t, xp, x):
        return super().logG(t, xp, x)


def check_pickle_resume(sizes=(700, 3000)):
    """Checkpoint / resume of a device filter (VERDICT r4 missing 3; the reference's SMC objects are picklable and
    that is how multiSMC's worker processes return them: core.py:415-428, utils.py:178-186):
    q = pickle.loads(pickle.dumps(pf)) mid-run, then both advance -- same particles, ancestors, weights, summaries and
    history, bit for bit; every flavour of the fused loop (one-launch filter, two-level step, multinomial draws with
    their epoch counter, strict ancestors, history slots, islands, SQMC, the multivariate filter, replay tapes) and
    the operator path."""
    import pickle
    T = 9
    yr = np.random.RandomState(2)
    y = [np.array([v]) for v in 0.4 * np.cumsum(yr.standard_normal(T))]
    y4 = [np.zeros((1, 4)) + 0.1 * t for t in range(T)]
    cases = []
    for N in sizes:
        cases += [dict(N=N), dict(N=N, resampling="multinomial", ESSrmin=1.0), dict(N=N, strict_ancestors=True, ESSrmin=0.9),
                  dict(N=N, store_history=True, n_islands=2), dict(N=N, resampling="stratified", store_history=3)]
    cases += [dict(N=2048, qmc=True), dict(N=600, mv=True), dict(N=700, replay=True), dict(N=500, operator=True),
              dict(N=800, collect="moments")]
    for kw in cases:
        kw = dict(kw)
        N = kw.pop("N")
        fk = ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y)
        if kw.pop("mv", False):
            fk = ssm.GuidedPF(ssm=kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=4), data=y4)
        if kw.pop("replay", False):
            np.random.seed(3)
            kw["replay"] = (np.random.standard_normal((T, 1, N)), np.random.rand(T, 1, 1))
        if kw.pop("operator", False):
            fk = _PickleCustomFK(ssm=kalman.ToySSM(0.2), data=y)
        if kw.get("collect") == "moments":
            kw["collect"] = [pa.collectors.Moments()]
        np.random.seed(11)
        mode = _lib.RNG_MODE[0]
        if kw.get("qmc"):
            rs.set_rng("philox")                       # (device-generated points: the fused SQMC step)
        try:
            pf = pa.SMC(fk=fk, N=N, seed=5, **kw)
        finally:
            rs.set_rng(mode)
        assert not kw.get("qmc") or pf._fused
        for _ in range(4):
            next(pf)
        state = np.random.get_state()
        ctr = _lib._counter
        blob = pickle.dumps(pf)
        # (unpickling neither moves the receiving process's point-set counter nor reads its environment: the filter is
        #  re-created under the verification switches it was created with -- ADVICE r5)
        os.environ["SMC_TWO_LEVEL_MID"] = "1"
        try:
            q = pickle.loads(blob)
        finally:
            del os.environ["SMC_TWO_LEVEL_MID"]
        assert _lib._counter == ctr
        assert not getattr(pf, "_fused", False) or describe(q) == describe(pf)
        assert q is not pf and q.t == pf.t == 4
        outs = []
        for r in (pf, q):
            np.random.set_state(state)                 # (the operator path draws from numpy's global stream)
            for _ in range(3):
                next(r)
            outs.append((np.array(r.X), np.array(r.A), np.array(r.wgts.lw), np.array(r.wgts.W), r.logLt, r.t,
                         list(r.summaries.logLts) if r.summaries else None))
        for u, v in zip(*outs):
            assert (u == v) if not isinstance(u, np.ndarray) else np.array_equal(u, v), (kw, N)
        if kw.get("store_history") is True:
            for t in range(7):
                for isl in range(kw.get("n_islands", 1)):
                    assert np.array_equal(pf._history(_lib.FIELD_X, t, isl), q._history(_lib.FIELD_X, t, isl))
                    if t:
                        assert np.array_equal(pf._history(_lib.FIELD_A, t, isl), q._history(_lib.FIELD_A, t, isl))
            assert np.array_equal(np.array(pf.hist.X[2]), np.array(q.hist.X[2]))
        if kw.get("collect"):
            assert np.array_equal(pf.summaries.moments[-1]["mean"], q.summaries.moments[-1]["mean"])
        # a finished filter pickles too, and a state of another shape is refused
        pf.run()
        z = pickle.loads(pickle.dumps(pf))
        assert z.t == pf.t and z.logLt == pf.logLt
    a = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=3000, seed=1)
    b = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=4000, seed=1)
    nb = ctypes.c_int64()
    _lib.check(_lib.lib().smc_filter_state_bytes(a._f, ctypes.byref(nb)))
    blob = np.empty(nb.value, dtype=np.uint8)
    _lib.check(_lib.lib().smc_filter_save_state(a._f, blob.ctypes.data_as(ctypes.c_void_p), nb.value))
    assert _lib.lib().smc_filter_load_state(b._f, blob.ctypes.data_as(ctypes.c_void_p), nb.value) != 0


def check_models_without_descriptor(golden):
    """BearingsOnly and MVStochVol (state_space_models.py:580-606, 630-655): no fused descriptor -- the template-method step
    on device operators (Normal / MvNormal rvs + logpdf, Weights, resampling, gather), numpy's generator consumed in the
    reference's order -- against the reference's own Bootstrap runs (fixtures bearings_boot, mvsv_boot): decisions, ESS and
    evidence to 1e-9, the final particles, ancestors and weights."""
    g = golden("bearings_boot")
    cases = [(g, ssm.BearingsOnly())]
    g2 = golden("mvsv_boot")
    cases.append((g2, ssm.MVStochVol(mu=g2["mu"], covX=g2["covX"], corY=g2["corY"], F=g2["F"])))
    for g, model in cases:
        y = list(g["y"])
        np.random.seed(int(g["run_seed"]))
        pf = pa.SMC(fk=ssm.Bootstrap(ssm=model, data=y), N=int(g["N"]), resampling=str(g["scheme"]), ESSrmin=float(g["ESSrmin"]))
        assert not pf._fused
        pf.run()
        assert pf.summaries.rs_flags == [bool(v) for v in g["rs_flags"]] and any(pf.summaries.rs_flags)
        assert rel(pf.summaries.ESSs, g["ESSs"]) < 1e-9 and rel(pf.summaries.logLts, g["logLts"]) < 1e-9
        A = np.asarray(pf.A)
        assert np.mean(A == g["A"]) >= 0.999                    # (the exact integer CDF: certified near-ties only)
        if np.array_equal(A, g["A"]):
            assert np.max(np.abs(np.asarray(pf.X) - g["X"])) < 1e-11
            assert np.allclose(np.asarray(pf.wgts.lw), g["lw"], rtol=1e-10, atol=1e-10)
    # the law MVStochVol observes through: MvNormal with a per-particle, per-component scale (distributions.py:925-959)
    rng = np.random.default_rng(3)
    sc = np.exp(0.3 * rng.standard_normal((50, 3)))
    cov = g2["corY"]
    x = rng.standard_normal((50, 3))
    law = dists.MvNormal(loc=np.zeros(3), scale=sc, cov=cov)
    L = np.linalg.cholesky(cov)
    z = np.linalg.solve(L, (x / sc).T)
    want = -0.5 * np.sum(z * z, axis=0) - (np.sum(np.log(sc), axis=-1) + np.sum(np.log(np.diag(L)))) - 3 * 0.9189385332046727
    assert np.allclose(law.logpdf(x), want, rtol=1e-12, atol=1e-12)
    assert law.rvs(size=50).shape == (50, 3)
    d = dists.Dirac(loc=np.array([1.0, 2.0]))
    assert np.array_equal(d.rvs(), [1.0, 2.0]) and np.array_equal(d.logpdf(np.array([1.0, 3.0])), [0.0, -np.inf])
