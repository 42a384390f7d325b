"""Time of the hand-written radix argsort (csrc/smc_sort.hip) on resident data.
    python tools/sort_perf.py            (on a GPU box)"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from particles_amd import _lib
from particles_amd._lib import DeviceArray, check, lib

for log2N in (12, 16, 20, 22):
    N = 1 << log2N
    x = DeviceArray.from_numpy(np.random.default_rng(1).standard_normal(N))
    out = DeviceArray((N,), dtype=np.int64)
    ctx = _lib.ctx()
    for rep in range(3):
        check(lib().smc_argsort(ctx.h, x.ptr, N, out.ptr))
    check(lib().smc_ctx_sync(ctx.h))
    t0 = time.perf_counter()
    R = 20
    for rep in range(R):
        check(lib().smc_argsort(ctx.h, x.ptr, N, out.ptr))
    check(lib().smc_ctx_sync(ctx.h))
    dt = (time.perf_counter() - t0) / R
    o = out.get()
    ok = np.array_equal(o, np.argsort(x.get(), kind="stable"))
    print("argsort N=2^%d: %.3f ms  %.1f M keys/s  %.1f GB/s of the 8 x 40 B passes  correct=%s"
          % (log2N, 1e3 * dt, N / dt / 1e6, 8 * 40.0 * N / dt / 1e9, ok))
