"""Where a step of a USER-DEFINED model (the template-method step on device operators) spends its time: host side by
cProfile, device side by `rocprofv3 --kernel-trace --stats -- python tools/generic_profile.py trace`.
    python tools/generic_profile.py [trace]"""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, particles_amd as pa
from particles_amd import distributions as dists, resampling as rs, state_space_models as ssm


class UserToySSM(ssm.StateSpaceModel):
    default_params = {"sigma": 0.2}

    def PX0(self):
        return dists.Normal()

    def PX(self, t, xp):
        return dists.Normal(loc=xp)

    def PY(self, t, xp, x):
        return dists.Normal(loc=x, scale=self.sigma)


N, K, W = 1 << 20, 60, 5
y = bench.synthetic_data(W + 2 * K + 2)
pa.set_resident(True); rs.set_rng("philox")
pf = pa.SMC(fk=ssm.Bootstrap(ssm=UserToySSM(sigma=0.2), data=y), N=N, resampling="systematic", ESSrmin=0.5, collect="off")
for _ in range(W):
    next(pf)
t0 = time.perf_counter()
for _ in range(K):
    next(pf)
ll = float(pf.logLt)
print("N = 2^20: %.1f us per step (%d steps), logLt %.4f" % (1e6 * (time.perf_counter() - t0) / K, K, ll), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "trace":
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    next(pf)
ll = float(pf.logLt)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
