"""us per step of the fused bootstrap filter (ToySSM, systematic, ESSrmin 0.5: resamples every step)
over population sizes that are and are not powers of two -- which step each lands on and what it
costs.    python tools/size_sweep.py        (on a GPU box)"""
import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import particles_amd as pa                                            # noqa: E402
from particles_amd import _lib, kalman, state_space_models as ssm     # noqa: E402
from bench import synthetic_data                                      # noqa: E402

T = 460
y = synthetic_data(T)
for N in (1000, 10 ** 4, 1 << 14, 10 ** 5, 1 << 17, 10 ** 6, 1 << 20, 3 * 10 ** 6, 1 << 22, 10 ** 7, 1 << 24):
    pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=N, seed=3, collect="off")
    buf = ctypes.create_string_buffer(256)
    _lib.check(_lib.lib().smc_filter_describe(pf._f, buf, 256))
    pf.step_async(60); pf.sync()
    t0 = time.perf_counter()
    pf.step_async(400); pf.sync()
    dt = (time.perf_counter() - t0) / 400
    print("N = %-9d %-42s %8.2f us/step  %7.2f G particle-steps/s" % (N, buf.value.decode(), 1e6 * dt, N / dt / 1e9),
          flush=True)
