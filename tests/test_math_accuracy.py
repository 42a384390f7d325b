"""Accuracy of the lean fp64 elementary functions (csrc/smc_math.h) against
numpy's libm, evaluated on the host through the emulator build."""
import ctypes
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


@pytest.fixture(scope="module")
def emu():
    import build_emu
    L = ctypes.CDLL(build_emu.build())
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def ulps(got, want):
    return np.abs(got - want) / np.spacing(np.abs(want))


def test_exp_nonpos(emu):
    rng = np.random.default_rng(0)
    x = np.concatenate([-rng.exponential(1.0, 200000), -rng.uniform(0, 745, 200000),
                        -10.0 ** rng.uniform(-300, -1, 50000),
                        np.array([0.0, -0.0, -1e-320, -708.0, -745.0, -745.13, -746.0, -1e4,
                                  -np.inf])])
    out = np.empty_like(x)
    emu.smc_test_exp_nonpos(_p(x), x.size, _p(out))
    want = np.exp(x)
    norm = want > 1e-300
    assert ulps(out[norm], want[norm]).max() <= 1.0
    assert np.all(out[~norm] >= 0) and np.allclose(out[~norm], want[~norm], atol=1e-300)
    assert out[-1] == 0.0 and out[0 - 9] == 1.0
    nan = np.array([np.nan])
    o = np.empty(1)
    emu.smc_test_exp_nonpos(_p(nan), 1, _p(o))
    assert np.isnan(o[0])


def test_log_pos(emu):
    rng = np.random.default_rng(2)
    x = np.concatenate([(rng.integers(0, 2 ** 52, 400000) + 0.5) * 2.0 ** -52,
                        10.0 ** rng.uniform(-300, 300, 100000), rng.uniform(0.5, 2.0, 200000),
                        np.array([2.0 ** -53, 1.0 - 2.0 ** -53, 1.0, 0.5, 2.0, 0.70710678118654752])])
    out = np.empty_like(x)
    emu.smc_test_log_pos(_p(x), x.size, _p(out))
    want = np.log(x.astype(np.longdouble)).astype(np.float64)
    nz = want != 0
    assert ulps(out[nz], want[nz]).max() <= 1.0
    assert np.all(out[~nz] == 0.0)


def test_sincospi_02(emu):
    rng = np.random.default_rng(1)
    a = np.concatenate([rng.uniform(0, 2, 400000), ((rng.integers(0, 2 ** 52, 100000) + 0.5) * 2.0 ** -52) * 2,
                        np.array([0.0, 0.25, 0.5, 0.75, 1.0, 1.25, 1.5, 1.75, 2.0, 2.0 ** -52, 1e-300])])
    s = np.empty_like(a)
    c = np.empty_like(a)
    emu.smc_test_sincospi_02(_p(a), a.size, _p(s), _p(c))
    al = a.astype(np.longdouble)
    ws = np.sin(np.pi * al.astype(np.longdouble) * (np.longdouble(np.pi) / np.longdouble(float(np.pi)))) if False else None
    # reference in extended precision: sin(pi a) with pi in long double
    pi_l = np.longdouble("3.14159265358979323846264338327950288")
    ws = np.sin(pi_l * al).astype(np.float64)
    wc = np.cos(pi_l * al).astype(np.float64)
    # absolute error relative to the unit circle (what Box-Muller needs) ...
    assert np.max(np.abs(s - ws)) < 2.3e-16 and np.max(np.abs(c - wc)) < 2.3e-16
    # ... and relative error away from the zeros
    big = np.abs(ws) > 1e-3
    assert ulps(s[big], ws[big]).max() <= 2.0
    big = np.abs(wc) > 1e-3
    assert ulps(c[big], wc[big]).max() <= 2.0
    assert s[-11] == 0.0 and c[-11] == 1.0 and s[-9] == 1.0 and c[-7] == -1.0


def test_box_muller_pair_table_driven(emu):
    """smc_bm_pair (table-driven log / sin / cos, rsq-based sqrt) on 2 x 52 random bits against the
    definition in extended precision: z = sqrt(-2 log u1) (cos, sin)(2 pi u2), u = (k + 1/2) 2^-52.
    Error budget: |dz| <= 6e-16 x radius (unit-circle error 2e-16 + radius 4 ulp); moments of N(0, 1)."""
    rng = np.random.default_rng(0)
    n = 1500000
    a = rng.integers(0, 2 ** 64, n, dtype=np.uint64)
    b = rng.integers(0, 2 ** 64, n, dtype=np.uint64)
    edge = np.array([0, (1 << 64) - 1, 1 << 12, 1 << 63, (1 << 63) - 1, (1 << 63) + (1 << 12)], dtype=np.uint64)
    a[:6] = edge
    b[6:12] = edge
    ne = 12 + 2000
    a[12:1012] = np.uint64((1 << 64) - 1) - (rng.integers(0, 2 ** 30, 1000, dtype=np.uint64) << np.uint64(12))   # u1 -> 1
    a[1012:2012] = rng.integers(0, 2 ** 20, 1000, dtype=np.uint64) << np.uint64(12)                              # u1 -> 0
    # every node boundary of the log table: mantissas at (2 j + 1) / 256 +- 1 ulp
    z0 = np.empty(n)
    z1 = np.empty(n)
    pu = ctypes.POINTER(ctypes.c_uint64)
    emu.smc_test_bm_pair(a.ctypes.data_as(pu), b.ctypes.data_as(pu), ctypes.c_int64(n), _p(z0), _p(z1))
    ld = np.longdouble
    u1 = ((a >> np.uint64(12)).astype(ld) + ld(0.5)) * ld(2.0) ** -52
    u2 = ((b >> np.uint64(12)).astype(ld) + ld(0.5)) * ld(2.0) ** -52
    pi_l = ld("3.14159265358979323846264338327950288")
    r = np.sqrt(-2 * np.log(u1))
    w0, w1 = r * np.cos(2 * pi_l * u2), r * np.sin(2 * pi_l * u2)
    rr = r.astype(np.float64)
    assert np.all(np.isfinite(z0)) and np.all(np.isfinite(z1))
    assert np.max(np.abs(z0 - w0).astype(np.float64) / rr) < 6e-16
    assert np.max(np.abs(z1 - w1).astype(np.float64) / rr) < 6e-16
    rad = np.sqrt(z0.astype(ld) ** 2 + z1.astype(ld) ** 2)
    assert float(np.max(np.abs(rad / r - 1))) < 5 * 2.0 ** -53
    zz = np.concatenate([z0[ne:], z1[ne:]])
    m = zz.size
    assert abs(zz.mean()) < 4 / np.sqrt(m) and abs(zz.var() - 1) < 4 * np.sqrt(2 / m)
    assert abs(np.mean(zz ** 3)) < 4 * np.sqrt(15 / m) and abs(np.mean(zz ** 4) - 3) < 4 * np.sqrt(96 / m)
    assert abs(np.mean(z0[ne:] * z1[ne:])) < 4 / np.sqrt(n)


def test_indep_prior_logpdf_matches_scipy():
    """smc2.IndepPrior.logpdf: closed forms in NumPy (no scipy import on the SMC^2 hot path) against
    scipy.stats, support boundaries included."""
    import numpy as np
    from scipy import stats
    from particles_amd import smc2
    rng = np.random.RandomState(0)
    p = smc2.IndepPrior(a=("normal", 0.3, 1.2), b=("lognormal", -0.5, 0.7), c=("uniform", -1.0, 2.0),
                        d=("gamma", 2.5, 1.5), e=("beta", 2.0, 3.5))
    th = p.rvs(1000, rng=rng)
    th["b"][:3] = [-1.0, 0.0, 1e-300]
    th["c"][:2] = [-2.0, 3.0]
    th["d"][:2] = [-1.0, 0.0]
    th["e"][:3] = [0.0, 1.0, 1.5]
    want = (stats.norm.logpdf(th["a"], 0.3, 1.2) + stats.lognorm.logpdf(th["b"], 0.7, scale=np.exp(-0.5))
            + stats.uniform.logpdf(th["c"], -1.0, 3.0) + stats.gamma.logpdf(th["d"], 2.5, scale=1 / 1.5)
            + stats.beta.logpdf(th["e"], 2.0, 3.5))
    got = p.logpdf(th)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin) and fin.sum() > 900
    assert np.max(np.abs(got[fin] - want[fin])) < 1e-13
