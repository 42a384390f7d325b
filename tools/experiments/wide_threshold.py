import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import particles_amd as pa
from particles_amd import _lib, kalman, state_space_models as ssm
from bench import synthetic_data
T = 1100
y = synthetic_data(T)
for N in (1 << 16, 1 << 17, 200000, 1 << 18, 300000, 1 << 19, 600000, 800000, 1000000, 1 << 20):
    res = []
    for rep in range(2):
        pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=N, seed=3, collect="off")
        buf = ctypes.create_string_buffer(256); _lib.check(_lib.lib().smc_filter_describe(pf._f, buf, 256))
        pf.step_async(60); pf.sync()
        t0 = time.perf_counter(); pf.step_async(1000); pf.sync()
        res.append(1e6 * (time.perf_counter() - t0) / 1000)
    print("N = %-9d %-30s %s" % (N, buf.value.decode(), " ".join("%.2f" % r for r in res)), flush=True)
