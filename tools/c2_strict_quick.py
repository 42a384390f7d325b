import sys, time
sys.path.insert(0, "/root/repo")
import bench, particles_amd as pa
from particles_amd import kalman, state_space_models as ssm
y = bench.synthetic_data(1200)
pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=1 << 20, seed=5, collect="off", strict_ancestors=True)
pf.step_async(100); pf.sync()
best = 1e9
for r in range(5):
    t0 = time.perf_counter(); pf.step_async(200); pf.sync(); best = min(best, (time.perf_counter() - t0) / 200)
print("C2 strict %.2f us/step" % (best * 1e6))
