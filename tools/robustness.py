import sys, time
sys.path.insert(0, ".")
import numpy as np, particles_amd as pa
from particles_amd import kalman, state_space_models as ssm
from bench import synthetic_data
y = synthetic_data(60)
for sig in (0.2, 1e-3, 1e-6, 1e-9):
    for N in (1 << 20, 10 ** 6, 1000):            # (10^6: the general counts, ragged last tile)
        pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(sig), data=y), N=N, seed=1)
        t0 = time.perf_counter(); pf.run(); dt = time.perf_counter() - t0
        ess = np.array(pf.summaries.ESSs)
        print("sigmaY=%g N=%d: %.2f ms/step, logLt=%.4g, min ESS=%.3g, NaN ESS steps=%d, unique ancestors last=%d"
              % (sig, N, dt / 60 * 1e3, pf.logLt, np.nanmin(ess), int(np.isnan(ess).sum()), np.unique(pf.A).size))
# data with a NaN observation and an outlier
y2 = [v.copy() for v in y]; y2[10] = np.array([np.nan]); y2[20] = np.array([1e6])
pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y2), N=1 << 16, seed=1); pf.run()
print("NaN / outlier data: logLt", pf.logLt, "ESS[10]", pf.summaries.ESSs[10], "ESS[20]", pf.summaries.ESSs[20])
# every scheme and both auxiliary variants on collapsing weights, N not a power of two
for scheme in ("systematic", "stratified", "multinomial"):
    pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(1e-7), data=y[:20]), N=300000, resampling=scheme, seed=2)
    pf.run()
    print("collapsing weights, N=300000, %s: logLt=%.6g unique ancestors last=%d" % (scheme, pf.logLt, np.unique(pf.A).size))
pf = pa.SMC(fk=ssm.AuxiliaryPF(ssm=ssm.StochVol(), data=[np.array([v]) for v in 0.5 * np.random.default_rng(1).standard_normal(40)]),
            N=10 ** 5, seed=3)
pf.run()
print("APF N=10^5 (two-level step): logLt=%.4f fused=%s" % (pf.logLt, pf._fused))
