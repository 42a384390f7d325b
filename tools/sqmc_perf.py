"""Time per step of SMC(qmc=True) (SQMC on device operators: Sobol' points, argsort / Hilbert
sort, inverse CDF, gathers, ppf moves), arrays resident in HBM, device-generated points.

    python tools/sqmc_perf.py            (on a GPU box)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import particles_amd as pa
from particles_amd import kalman, resampling as rs
from particles_amd import state_space_models as ssm


def run(model, N, T, label, fused=True, graph=False):
    np.random.seed(42)
    rs.set_rng("numpy")
    pa.set_resident(False)
    x, y = model.simulate(T)
    rs.set_rng("philox")
    pa.set_resident(True)
    from particles_amd import _lib
    _lib.FUSED_SQMC[0] = fused
    try:
        out = []
        for rep in range(3):
            pa.seed(7 + rep)
            pf = pa.SMC(fk=ssm.Bootstrap(ssm=model, data=y), N=N, qmc=True, collect="off", use_graph=graph)
            t0 = time.perf_counter()
            pf.run()
            ll = float(pf.logLt)              # forces completion
            out.append((time.perf_counter() - t0) / T)
        print("%-28s N=2^%-2d  %-11s %8.3f ms/step  %7.2f M particle-steps/s   logLt %.3f"
              % (label, int(np.log2(N)), ("fused+graph" if graph else "fused") if pf._fused else "operators", 1e3 * min(out), N / min(out) / 1e6, ll))
    finally:
        pa.set_resident(False)
        rs.set_rng("numpy")
        _lib.FUSED_SQMC[0] = True


if __name__ == "__main__":
    if len(sys.argv) > 1:                       # python tools/sqmc_perf.py 20 [T]: one size, for profiling
        run(kalman.ToySSM(0.2), 1 << int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 30, "ToySSM d=1")
        sys.exit(0)
    for k in (7, 10):                           # below two tiles: the flat step (eager launches)
        run(kalman.ToySSM(0.2), 1 << k, 100, "ToySSM d=1")
        run(kalman.ToySSM(0.2), 1 << k, 100, "ToySSM d=1", fused=False)
    for k in (12, 13, 14, 16, 20, 22):
        run(kalman.ToySSM(0.2), 1 << k, 30, "ToySSM d=1")
        run(kalman.ToySSM(0.2), 1 << k, 30, "ToySSM d=1", graph=True)
        run(kalman.ToySSM(0.2), 1 << k, 30, "ToySSM d=1", fused=False)
    for k in (12, 16, 18, 20):
        run(kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=2), 1 << k, 20, "MVLinearGauss d=2 (Hilbert)")
        run(kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=2), 1 << k, 20, "MVLinearGauss d=2 (Hilbert)", fused=False)
