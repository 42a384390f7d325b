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

# ---- one SQMC step (core.py:339-349) on device operators, points generated on the device:
# with the closed-form sorted Sobol' order (rqmc.sobol_sorted) and with the generic argsort
import particles_amd as pa
from particles_amd import kalman, rqmc, resampling as rs, state_space_models as ssm
rs.set_rng("philox")
pa.set_resident(True)
T = 12
y = [np.array([v]) for v in np.random.default_rng(2).standard_normal(T)]
for log2N in (16, 20):
    N = 1 << log2N
    for closed in (True, False):
        keep = rqmc.sobol_sorted
        if not closed:
            rqmc.sobol_sorted = lambda N, d: None
        try:
            pa.seed(3)
            pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=N, qmc=True)
            next(pf); next(pf)
            check(lib().smc_ctx_sync(_lib.ctx().h))
            t0 = time.perf_counter()
            for _ in range(T - 2):
                next(pf)
            check(lib().smc_ctx_sync(_lib.ctx().h))
            dt = (time.perf_counter() - t0) / (T - 2)
            print("SQMC step N=2^%d (%s): %.3f ms  %.1f M particle-steps/s  logLt %.6f"
                  % (log2N, "closed-form Sobol' order" if closed else "argsort of the first coordinate",
                     1e3 * dt, N / dt / 1e6, pf.logLt))
        finally:
            rqmc.sobol_sorted = keep
