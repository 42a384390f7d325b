import ctypes, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import particles_amd as pa
from particles_amd import _lib, kalman, state_space_models as ssm
L = _lib.lib()
for log2N in (16, 18, 20, 22):
    N = 1 << log2N
    y = [np.array([0.1 * t]) for t in range(40)]
    pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=N, collect="off", seed=1, use_graph=False)
    pf.step_async(10); pf.sync()
    ctx = _lib.ctx()
    for _ in range(5): L.smc_filter_step(pf._f, 0)
    ctx.sync()
    L.smc_timer_start(ctx.h)
    R = 300
    for _ in range(R): L.smc_filter_step(pf._f, 0)
    ms = ctypes.c_float(); L.smc_timer_stop(ctx.h, ctypes.byref(ms))
    print("N=2^%d finalize-only k_prepare: %.2f us per launch" % (log2N, ms.value * 1e3 / R))
