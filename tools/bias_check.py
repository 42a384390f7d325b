"""Is the device's log-evidence estimator centred where the reference's is?  C2's model (ToySSM, T = 1000,
systematic, ESSrmin 0.5): R independent runs of the device filter at N = 2^14 and 2^20 and of the oracle
(the NumPy restatement of particles.SMC, pinned to the reference) at N = 2^14, against the exact Kalman
log-likelihood.  E[log L_hat] = log L - var / 2 + O(skew): both sides must show the same offset.
    python tools/bias_check.py [R]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                        # noqa: E402
import particles_amd as pa                                          # noqa: E402
from particles_amd import kalman, state_space_models as ssm         # noqa: E402
from oracle import smc_oracle as orc                                # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = 1000
y = bench.synthetic_data(T)
ll, _ = orc.kalman_loglik(orc.ToySSM(0.2), [np.atleast_1d(v) for v in np.squeeze(np.array(y))])
fk = ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y)


def report(name, v):
    v = np.array(v) - ll
    sd = v.std(ddof=1)
    print("%-28s n=%3d  mean %+8.4f  se %.4f  sd %.4f  -var/2 %+.4f  z(mean + var/2) %+.2f  min %+.3f max %+.3f"
          % (name, len(v), v.mean(), sd / np.sqrt(len(v)), sd, -0.5 * sd * sd,
             (v.mean() + 0.5 * sd * sd) / (sd / np.sqrt(len(v))), v.min(), v.max()), flush=True)


for log2N in (14, 20):
    out = []
    for s in range(R):
        pf = pa.SMC(fk=fk, N=1 << log2N, seed=1000 + s, collect="off")
        pf.run()
        out.append(pf.logLt)
    report("device N=2^%d" % log2N, out)
out = []
for s in range(R):
    np.random.seed(5000 + s)
    out.append(orc.run_filter(orc.ToySSM(0.2), y, 1 << 14, "systematic", 0.5)["final_logLt"])
report("oracle (reference) N=2^14", out)
