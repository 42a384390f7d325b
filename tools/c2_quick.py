"""C2 (and C3 systematic) per-step time, default and strict: a quick A/B probe for the GPU box.  python tools/c2_quick.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, particles_amd as pa
from particles_amd import kalman, state_space_models as ssm
y = bench.synthetic_data(1500)
for name, model, N, strict, ess in (("C2", kalman.ToySSM(0.2), 1 << 20, False, 0.5), ("C2 strict", kalman.ToySSM(0.2), 1 << 20, True, 0.5),
                                    ("C3 systematic", ssm.StochVol(), 1 << 22, False, 1.0)):
    pf = pa.SMC(fk=ssm.Bootstrap(ssm=model, data=y), N=N, seed=5, collect="off", strict_ancestors=strict, ESSrmin=ess)
    pf.step_async(100); pf.sync()
    best = 1e9
    for r in range(5):
        t0 = time.perf_counter(); pf.step_async(250); pf.sync(); best = min(best, (time.perf_counter() - t0) / 250)
    print("%-14s %.2f us/step  %.2f G particle-steps/s" % (name, best * 1e6, N / best / 1e9), flush=True)
    del pf
