"""Randomised check of the radix sort's four-pass form (csrc/smc_sort.hip) on the GPU: sizes 8 193 .. 3 000 000 (not
powers of two), data shapes that move the window and populate the groups the fix-up orders, weighted quantiles through the
same sort -- against np.argsort(kind="stable").    python tools/sort_fuzz.py [cases]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from particles_amd import _lib, hilbert, resampling as rs

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(77)
_lib.check(_lib.lib().smc_debug_sort_window_min(8193))
t0 = time.time()
kinds = {}
for c in range(ncases):
    N = int(rng.integers(8193, 3_000_000)) if c % 4 else int(rng.integers(8193, 40000))
    kind = c % 8
    if kind == 0:
        x = rng.standard_normal(N) * 10.0 ** rng.integers(-200, 200)
    elif kind == 1:
        x = rng.uniform(-1, 1) * 10.0 ** rng.integers(-5, 8) + 10.0 ** rng.integers(-12, 2) * rng.standard_normal(N)
    elif kind == 2:
        x = rng.integers(-50, 50, size=N).astype(np.float64) * rng.random()
    elif kind == 3:                                   # children of a few parents, jitter far below the parents' spread
        npar = int(rng.integers(1, 40))
        x = rng.standard_normal(npar)[rng.integers(0, npar, size=N)]
        x = x + 10.0 ** rng.integers(-16, -6) * rng.standard_normal(N)
    elif kind == 4:                                   # neighbours a few ulps apart
        b = rng.standard_normal(N)
        idx = rng.integers(0, N, size=N // 2)
        b[idx] = np.nextafter(b[(idx * 7 + 1) % N], np.inf)
        x = b
    elif kind == 5:
        x = np.exp(rng.uniform(-700, 700, size=N)) * rng.choice([-1.0, 1.0], size=N)
    elif kind == 6:                                   # dense cluster of distinct values + background: the fallback
        x = rng.standard_normal(N)
        m = int(rng.integers(33, 3000))
        x[rng.choice(N, m, replace=False)] = 0.25 + rng.permutation(m) * 2.0 ** -54
    else:
        x = np.round(rng.standard_normal(N), int(rng.integers(0, 6)))
    o = np.asarray(hilbert.argsort(x))
    ref = np.argsort(x, kind="stable")
    nz = x[ref] != 0.0
    assert np.array_equal(x[o], x[ref]) and np.array_equal(o[nz], ref[nz]), (c, N, kind, int(np.sum(o != ref)))
    if c % 10 == 0:                                   # wquantiles: the same sort with the weights as payload
        W = rng.random(N); W /= W.sum()
        q = rs.wquantiles(W, x, alphas=(0.1, 0.5, 0.9))
        cw = np.cumsum(W[ref])
        for a, v in zip((0.1, 0.5, 0.9), np.atleast_1d(q)):
            k = int(np.searchsorted(cw, a))
            lo_, hi_ = x[ref][max(k - 2, 0)], x[ref][min(k + 2, N - 1)]
            assert lo_ <= v <= hi_, (c, a, v, lo_, hi_)
    kinds[kind] = kinds.get(kind, 0) + 1
print("sort fuzz: %d cases (sizes 8193 .. 3e6, 8 data shapes), every permutation == np.argsort(kind='stable'); %.0f s" % (ncases, time.time() - t0))
