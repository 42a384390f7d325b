"""Micro-benchmarks of the stand-alone device operators (run on the GPU box)."""
import ctypes
import sys
import os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import particles_amd as pa
from particles_amd import _lib
from particles_amd._lib import lib, check, DeviceArray

N = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
REPS = 200
ctx = _lib.ctx()
L = lib()


def timeit(name, fn, bytes_moved):
    for _ in range(5):
        fn()
    ctx.sync()
    check(L.smc_timer_start(ctx.h))
    for _ in range(REPS):
        fn()
    ms = ctypes.c_float()
    check(L.smc_timer_stop(ctx.h, ctypes.byref(ms)))
    us = ms.value * 1e3 / REPS
    print("%-28s %8.2f us   %8.1f GB/s" % (name, us, bytes_moved / us / 1e3))


rng = np.random.default_rng(0)
lw = DeviceArray.from_numpy(rng.standard_normal(N))
W = DeviceArray((N,))
x = DeviceArray.from_numpy(rng.standard_normal(N))
out = DeviceArray((N,))
A = DeviceArray((N,), np.int64)
one = DeviceArray.from_numpy(np.array([0.7]))
zero = DeviceArray.from_numpy(np.array([0.1]))
o4 = (ctypes.c_double * 4)()
check(L.smc_lse_normalise(ctx.h, lw.ptr, N, W.ptr, o4))

timeit("standard_normal (philox+BM)", lambda: L.smc_standard_normal(ctx.h, 3, N, out.ptr), 8 * N)
timeit("uniform (philox)", lambda: L.smc_uniform(ctx.h, 3, N, out.ptr), 8 * N)
timeit("normal_logpdf", lambda: L.smc_normal_logpdf(ctx.h, x.ptr, 1, zero.ptr, 0, one.ptr, 0, N, out.ptr), 16 * N)
timeit("normal_rvs replay", lambda: L.smc_normal_rvs(ctx.h, x.ptr, 1, one.ptr, 0, lw.ptr, 0, N, out.ptr), 24 * N)
timeit("normal_rvs philox", lambda: L.smc_normal_rvs(ctx.h, x.ptr, 1, one.ptr, 0, None, 0, N, out.ptr), 16 * N)
timeit("gather", lambda: L.smc_gather(ctx.h, x.ptr, A.ptr, N, 1, out.ptr), 24 * N)
timeit("d2d copy 8N", lambda: L.smc_memcpy_d2d(ctx.h, out.ptr, x.ptr, 8 * N), 16 * N)
u = DeviceArray.from_numpy(np.array([0.3]))
timeit("resample systematic", lambda: L.smc_resample(ctx.h, 2, W.ptr, N, N, u.ptr, 0, A.ptr), 24 * N)
timeit("resample stratified philox", lambda: L.smc_resample(ctx.h, 1, W.ptr, N, N, None, 0, A.ptr), 24 * N)

# ---- PCIe-inclusive rate of the host-facing (numpy in -> numpy out) operators
import time
from particles_amd import resampling as rs
Wh = W.get()
lwh = lw.get()
for name, fn, nbytes in (
        ("rs.systematic host->host", lambda: rs.systematic(Wh, N), 16 * N),
        ("rs.Weights host->host", lambda: rs.Weights(lw=lwh.copy()).W, 16 * N)):
    fn()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    dt = (time.perf_counter() - t0) / 20
    print("%-28s %8.1f us   %8.2f GB/s over PCIe (H2D + D2H incl. numpy-side copies)" % (name, dt * 1e6, nbytes / dt / 1e9))
