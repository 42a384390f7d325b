"""Soak of the strict-ancestors path (csrc/smc_seqx.h) on the GPU: (1) the two-launch emulation of the reference's
sequential fp64 prefix sums against the tile walk (smc_seq_prefix_sums mode 0 vs 2: two independent parallel forms of the
same definition) on a few thousand random weight vectors of every shape we could think of, sizes around tile edges;
(2) inverse_cdf(strict) against a bisection of those sums; (3) long strict filter runs (C2's model, islands, the three
schemes), every resampling step's statistics read back: exceptions walked, exact-path steps.
    python tools/strict_soak.py [cases]"""
import ctypes
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                        # noqa: E402
import particles_amd as pa                                          # noqa: E402
from particles_amd import _lib, kalman, resampling as rs, state_space_models as ssm   # noqa: E402
from particles_amd._lib import DeviceArray, check, lib, ctx         # noqa: E402

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = np.random.default_rng(2026)


def seq(d, S, N, mode):
    c = ctypes.c_int64(-9)
    check(lib().smc_seq_prefix_sums(ctx().h, d.ptr, N, S.ptr, mode, ctypes.byref(c)))
    return c.value


def weights(N, kind):
    if kind == 0:
        w = np.exp(rng.uniform(0.1, 60.0) * rng.standard_normal(N))
    elif kind == 1:
        w = rng.random(N) * (rng.random(N) > rng.uniform(0.0, 0.99))
    elif kind == 2:                                   # a few heavy particles, the rest negligible
        w = np.exp(rng.uniform(-700.0, -5.0) + rng.standard_normal(N))
        w[rng.choice(N, rng.integers(1, 20), replace=False)] = rng.random() + 0.1
    elif kind == 3:                                   # dyadic / exactly summable
        w = rng.integers(0, 1 << rng.integers(1, 30), size=N).astype(np.float64)
    elif kind == 4:                                   # piecewise constant blocks of very different scales
        w = np.repeat(np.exp(30.0 * rng.standard_normal(-(-N // 777))), 777)[:N]
    elif kind == 5:                                   # a Gaussian likelihood over a sorted state (what a filter sees)
        x = np.sort(rng.standard_normal(N)) * rng.uniform(0.5, 30.0)
        w = np.exp(-0.5 * ((x - rng.standard_normal()) / rng.uniform(0.01, 2.0)) ** 2)
    else:                                             # unnormalised, huge or tiny overall scale
        w = rng.random(N) * 10.0 ** rng.integers(-300, 300)
    s = w.sum()
    return w / s if (kind != 6 and s > 0 and np.isfinite(s)) else w


t0 = time.time()
exact = 0
worst = 0
KINDS = ("lognormal", "sparse uniform", "few heavy", "integers", "constant blocks", "gaussian over sorted", "unnormalised 1e+-300")
WHY = {1: "lists full", 2: "walk left its binade", 4: "segment binade", 8: "head binade", 16: "?"}
by_kind = {}
sizes = [1025, 2048, 3000, 1 << 14, (1 << 16) + 1, 1 << 18, 1 << 20, (1 << 20) + 1023, 1 << 22]
for c in range(ncases):
    N = sizes[c % len(sizes)] if c % 7 else int(rng.integers(1025, 300000))
    W = weights(N, c % 7)
    if not np.all(np.isfinite(W)) or W.sum() <= 0:
        continue
    d = DeviceArray.from_numpy(W)
    Sa, Sb = DeviceArray((N,)), DeviceArray((N,))
    fb = seq(d, Sa, N, 0)
    seq(d, Sb, N, 2)
    a, b = Sa.get(), Sb.get()
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), (c, N, c % 7, int((a != b).sum()))
    exact += fb < 0
    worst = max(worst, fb)
    rec = by_kind.setdefault(c % 7, [0, {}])
    rec[0] += 1
    if fb < 0:
        rec[1][-fb] = rec[1].get(-fb, 0) + 1
    if c % 5 == 0:                                    # the searches against it
        M = N if c % 10 else N // 3 + 7
        su = np.sort(rng.random(M)) * min(1.0, a[-1])
        got = np.asarray(rs.inverse_cdf(su, W, strict=True))
        want = np.minimum(np.searchsorted(a, su, side="left"), N - 1)
        assert np.array_equal(got, want), (c, N, int((got != want).sum()))
print("operator: %d weight vectors, two-launch form == tile walk bit for bit; exact path taken %d times, most exceptions "
      "walked %d; %.0f s" % (ncases, exact, worst, time.time() - t0), flush=True)
for k in sorted(by_kind):
    n, why = by_kind[k]
    print("    %-22s %4d vectors, exact path: %s" % (KINDS[k], n, ", ".join(
        "%d x (%s)" % (v, " + ".join(WHY[b] for b in WHY if r & b)) for r, v in sorted(why.items())) or "never"), flush=True)

y = bench.synthetic_data(1200)
for scheme, N, nisl in (("systematic", 1 << 20, 1), ("stratified", 1 << 18, 3), ("multinomial", (1 << 19) + 4321, 2), ("systematic", 5000, 4),
                        ("systematic", 5000, 4), ("multinomial", 3000, 8), ("stratified", 70000, 2)):
    pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=N, seed=int(rng.integers(1, 1 << 30)), collect="off",
                strict_ancestors=True, resampling=scheme, n_islands=nisl)
    nx, ex, why = [], 0, {}
    every = 25 if N > 100000 else 1
    for k in range(0, 1200, every):
        pf.step_async(every)
        for isl in range(nisl):
            e, n = ctypes.c_int64(), ctypes.c_int64()
            check(lib().smc_filter_strict_stats(pf._f, isl, ctypes.byref(e), ctypes.byref(n)))
            ex += e.value != 0
            if e.value:
                why[(e.value, n.value)] = why.get((e.value, n.value), 0) + 1
            nx.append(n.value)
    assert np.all(np.isfinite(pf.logLts_islands))
    print("filter %-11s N=%-8d islands=%d: 1200 steps, exceptions per step %d .. %d (median %d), exact-path steps among the %d sampled: %d; "
          "logLt %s" % (scheme, N, nisl, min(nx), max(nx), int(np.median(nx)), len(nx), ex, np.round(pf.logLts_islands, 3)), flush=True)
    if why:
        print("    (why, exceptions): count  %s" % why, flush=True)
print("strict soak OK")
