"""Upper bound of moving the normals out of k_propagate: C2 with the step's normals read from a tape (replay mode:
no Philox / Box-Muller in the kernel, 8 B per particle more to read) against the production kernel."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, particles_amd as pa
from particles_amd import kalman, state_space_models as ssm
T, N = 140, 1 << 20
y = bench.synthetic_data(T)
fk = ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y)
rng = np.random.default_rng(1)
z = rng.standard_normal((T, 1, N)); u = rng.random((T, 1, 1))
for name, kw in (("philox", {}), ("tape", {"replay": (z, u)})):
    pf = pa.SMC(fk=fk, N=N, seed=5, collect="off", **kw)
    pf.step_async(20); pf.sync()
    ts = []
    for r in range(5):
        t0 = time.perf_counter(); pf.step_async(20); pf.sync(); ts.append((time.perf_counter() - t0) / 20)
    print(name, "%.2f us per step (K = 20 regions: %s)" % (1e6 * np.median(ts), " ".join("%.2f" % (1e6 * v) for v in ts)))
    del pf
