"""The radix sort's forms (eight passes / four passes over the 32-bit window + fix-up: csrc/smc_sort.hip; FORMS takes a
third entry for an experimental build, see profiles/r15_sort_onesweep_perf.txt), data shapes that stress the window, and
the fused SQMC step on each, timed.
    python tools/sort_quick.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, particles_amd as pa
from particles_amd import _lib, hilbert, kalman, resampling as rs, state_space_models as ssm
from particles_amd._lib import DeviceArray

rng = np.random.default_rng(3)
FORMS = (("eight passes", 1 << 40, 0), ("four + fix-up", 8193, 0))


def shapes(N):
    yield "normal", rng.standard_normal(N)
    x = rng.standard_normal(N); x[N // 3] = 3000.0
    yield "outlier 3e3 sd", x
    x = rng.standard_normal(N); x[N // 3] = 1e9
    yield "outlier 1e9 sd", x
    yield "cauchy", rng.standard_cauchy(N)
    yield "student t3", rng.standard_t(3, N)
    yield "lognormal s=3", np.exp(3.0 * rng.standard_normal(N))
    yield "lognormal s=20", np.exp(20.0 * rng.standard_normal(N))
    yield "10^U(-300,300)", rng.standard_normal(N) * 10.0 ** rng.integers(-300, 300, size=N)
    yield "1000 + 1e-3 z", 1000.0 + 1e-3 * rng.standard_normal(N)
    yield "uniform", rng.random(N)
    yield "sorted normal", np.sort(rng.standard_normal(N))
    par = np.repeat(rng.standard_normal(N // 8), 8)
    yield "offspring + 1e-3 z", par + 1e-3 * rng.standard_normal(N)
    yield "offspring + 1e-9 z", par + 1e-9 * rng.standard_normal(N)


for log2N in (20, 17):
    N = 1 << log2N
    for sname, xs in shapes(N):
        x = DeviceArray.from_numpy(xs)
        ref = np.argsort(xs, kind="stable")
        line = "N=2^%d %-20s" % (log2N, sname)
        for name, wm, lm in FORMS[1:]:
            _lib.check(_lib.lib().smc_debug_sort_window_min(wm))
                o = hilbert.argsort(x)
            ok = np.array_equal(np.asarray(o), ref)
            best = 1e9
            for r in range(3):
                t0 = time.perf_counter()
                for k in range(10):
                    o = hilbert.argsort(x)
                np.asarray(o)[:1]
                best = min(best, (time.perf_counter() - t0) / 10)
            line += "  %s %8.1f us%s" % (name, best * 1e6, "" if ok else " WRONG")
        print(line, flush=True)
for log2N in (14, 15, 16, 17, 18, 20, 22):
    N = 1 << log2N
    x = DeviceArray.from_numpy(rng.standard_normal(N))
    for name, wm, lm in FORMS:
        _lib.check(_lib.lib().smc_debug_sort_window_min(wm))
        o = hilbert.argsort(x)
        _lib.ctx_sync() if hasattr(_lib, "ctx_sync") else None
        best = 1e9
        for r in range(5):
            t0 = time.perf_counter()
            for k in range(20):
                o = hilbert.argsort(x)
            np.asarray(o)[:1]
            best = min(best, (time.perf_counter() - t0) / 20)
        print("argsort N=2^%d %-14s %8.1f us" % (log2N, name, best * 1e6), flush=True)
y = bench.synthetic_data(400)
for name, wm, lm in FORMS * 2:
    _lib.check(_lib.lib().smc_debug_sort_window_min(wm))
    rs.set_rng("philox")                                   # (device-generated points: the fused SQMC step)
    pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=1 << 20, seed=5, collect="off", qmc=True)
    assert pf._fused
    pf.step_async(50); pf.sync()
    best = 1e9
    for r in range(4):
        t0 = time.perf_counter(); pf.step_async(60); pf.sync(); best = min(best, (time.perf_counter() - t0) / 60)
    print("SQMC step N=2^20 %-14s %8.1f us  logLt %.6f" % (name, best * 1e6, pf.logLt), flush=True)
