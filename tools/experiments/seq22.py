import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from particles_amd._lib import DeviceArray, check, lib, ctx
rng = np.random.default_rng(1)
N = 1 << 22
for name, w in (("lognormal", np.exp(3 * rng.standard_normal(N))), ("uniform", np.ones(N)), ("zeros", rng.random(N) * (rng.random(N) > 0.3))):
    W = w / w.sum(); d = DeviceArray.from_numpy(W); S = DeviceArray((N,))
    for rep in range(3):
        c = ctypes.c_int64(-9)
        ctx().sync(); t0 = time.perf_counter()
        check(lib().smc_seq_prefix_sums(ctx().h, d.ptr, N, S.ptr, 0, ctypes.byref(c)))
        ctx().sync(); dt = time.perf_counter() - t0
        print(name, rep, "fallback", c.value, "%.1f us (sync per call)" % (dt * 1e6), flush=True)
    for reps in (1, 2, 5, 20):
        ctx().sync(); t0 = time.perf_counter()
        for _ in range(reps):
            check(lib().smc_seq_prefix_sums(ctx().h, d.ptr, N, S.ptr, 0, None))
        t1 = time.perf_counter(); ctx().sync(); t2 = time.perf_counter()
        print(name, "back to back x%d: enqueue %.1f us per call, total %.1f us per call" % (reps, (t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6), flush=True)
