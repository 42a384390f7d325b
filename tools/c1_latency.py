import sys, time
sys.path.insert(0, ".")
import numpy as np
import particles_amd as pa
from particles_amd import kalman, state_space_models as ssm
from bench import synthetic_data
y = synthetic_data(200)
fk = ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y)
pa.SMC(fk=fk, N=1000, seed=0).run()
for N in (1000, 100000):
    ts = []
    for r in range(5):
        t0 = time.perf_counter(); pf = pa.SMC(fk=fk, N=N, seed=r); t1 = time.perf_counter(); pf.run(); t2 = time.perf_counter(); ts.append((t1 - t0, t2 - t1))
    print("C1-like N=%d T=200: create %.2f ms, run %.2f ms (%.1f us/step)" % (N, 1e3 * np.median([a for a, _ in ts]), 1e3 * np.median([b for _, b in ts]), 1e6 * np.median([b for _, b in ts]) / 200))
