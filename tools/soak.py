import sys, time
sys.path.insert(0, ".")
import numpy as np
import particles_amd as pa
from particles_amd import kalman, state_space_models as ssm, _lib
from bench import synthetic_data
y = synthetic_data(30000)
pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=1 << 20, seed=1, collect="off")
t0 = time.perf_counter(); pf.step_async(30000); pf.sync(); dt = time.perf_counter() - t0
print("30000 steps: %.1f us/step, logLt %.3f finite=%s" % (dt / 30000 * 1e6, pf.logLt, np.isfinite(pf.logLt)))
from oracle import smc_oracle as orc
ll, _ = orc.kalman_loglik(orc.ToySSM(0.2), [np.atleast_1d(v) for v in y])
print("kalman", ll, "diff", pf.logLt - ll)
del pf
info0 = _lib.ctx().device_info()
for i in range(300):
    p = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y[:40]), N=1 << 18, seed=i, collect="off", n_islands=4)
    p.run(); v = p.logLt
    del p
print("300 create/run/destroy cycles ok", v)
