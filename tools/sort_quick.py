"""The radix sort's two forms (eight passes / four passes + fix-up: csrc/smc_sort.hip) and the fused SQMC step, timed.
    python tools/sort_quick.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, particles_amd as pa
from particles_amd import _lib, hilbert, kalman, resampling as rs, state_space_models as ssm
from particles_amd._lib import DeviceArray

rng = np.random.default_rng(3)
for log2N in (14, 15, 16, 17, 18, 20, 22):
    N = 1 << log2N
    x = DeviceArray.from_numpy(rng.standard_normal(N))
    for name, wm in (("eight passes", 1 << 40), ("four + fix-up", 8193)):
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
for name, wm in (("eight passes", 1 << 40), ("four + fix-up", 8193)):
    _lib.check(_lib.lib().smc_debug_sort_window_min(wm))
    rs.set_rng("philox")                                   # (device-generated points: the fused SQMC step)
    pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=1 << 20, seed=5, collect="off", qmc=True)
    assert pf._fused
    pf.step_async(50); pf.sync()
    best = 1e9
    for r in range(4):
        t0 = time.perf_counter(); pf.step_async(60); pf.sync(); best = min(best, (time.perf_counter() - t0) / 60)
    print("SQMC step N=2^20 %-14s %8.1f us  logLt %.6f" % (name, best * 1e6, pf.logLt), flush=True)
