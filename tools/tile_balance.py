"""How evenly do the offspring spread over the 1024-parent tiles of k_ancestors?
(perf diagnostics: the slowest tile sets the kernel's duration)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import particles_amd as pa
from particles_amd import kalman, state_space_models as ssm
from bench import synthetic_data

log2N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = 1 << log2N
y = synthetic_data(200)
pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=N, seed=123)
for t in (5, 50, 150):
    pf.step_async(t - pf.t)
    A = pf.A
    cnt = np.bincount(A // 1024, minlength=N // 1024)
    passes = (cnt + 1023) // 1024
    ess = pf.summaries and None
    print("t=%d tiles=%d offspring/tile: mean %.0f max %d p99 %d zero-tiles %d  passes: max %d mean %.2f  ESS/N %.3f"
          % (t, N // 1024, cnt.mean(), cnt.max(), np.percentile(cnt, 99), (cnt == 0).sum(), passes.max(),
             passes.mean(), pf._summ()[0, -1, 0] / N))
