import sys, time
sys.path.insert(0, ".")
import numpy as np
import particles_amd as pa
from particles_amd import distributions as dists, state_space_models as ssm, resampling as rs
from bench import synthetic_data

class ToySSM(ssm.StateSpaceModel):
    def PX0(self): return dists.Normal()
    def PX(self, t, xp): return dists.Normal(loc=xp)
    def PY(self, t, xp, x): return dists.Normal(loc=x, scale=self.sigma)

y = synthetic_data(60)
N = 1 << 20
for resident, mode in ((False, "numpy"), (True, "numpy"), (True, "philox")):
    pa.set_resident(resident); rs.set_rng(mode)
    pf = pa.SMC(fk=ssm.Bootstrap(ssm=ToySSM(sigma=0.2), data=y), N=N, collect="off")
    next(pf); next(pf)
    t0 = time.perf_counter()
    for _ in range(20): next(pf)
    dt = (time.perf_counter() - t0) / 20
    print("user-defined model, resident=%s rng=%s: %.2f ms/step = %.3f G particle-steps/s" % (resident, mode, dt * 1e3, N / dt / 1e9))
