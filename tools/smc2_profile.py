"""Where the host side of an SMC^2 resample-move event goes (run on the GPU box):
batch creation (Python model objects / smc_filter_create), the candidates' run from 0 to t,
theta-resampling of whole filters, acceptance copies.    python tools/smc2_profile.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from particles_amd import kalman, smc2, state_space_models as ssm   # noqa: E402
from particles_amd.core import SMC                                   # noqa: E402

Ntheta, Nx, T = 512, 512, 100
rng = np.random.default_rng(1)
x = np.cumsum(rng.standard_normal(T))
y = [np.array([v]) for v in x + 0.3 * rng.standard_normal(T)]
prior = smc2.IndepPrior(sigmaY=("lognormal", np.log(0.5), 0.5))
mk = lambda sigmaY: kalman.LinearGauss(rho=1.0, sigmaX=1.0, sigmaY=sigmaY, sigma0=1.0)
alg = smc2.SMC2(ssm_cls=mk, prior=prior, data=y, init_Nx=Nx, N=Ntheta, seed=1, nmcmc=2)


def timed(label, fn, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    dt = (time.perf_counter() - t0) / reps
    print("%-46s %8.2f ms" % (label, 1e3 * dt))
    return out


timed("python model objects (N_theta Feynman-Kac)", lambda: [ssm.Bootstrap(ssm=mk(float(s)), data=y)
                                                            for s in alg.theta["sigmaY"]])
cand = timed("batch creation (_batch: objects + create)", lambda: alg._batch(alg.theta, Nx))
def run_to(t):
    c = alg._batch(alg.theta, Nx)
    c.step_async(t)
    return c.logLts_islands
timed("batch creation + run 0..50 + evidences", lambda: run_to(50))
timed("batch creation + run 0..100 + evidences", lambda: run_to(100))
alg.pf.step_async(50)
alg.pf.sync()
A = np.sort(rng.integers(0, Ntheta, Ntheta))
from particles_amd._lib import check, lib
timed("permute_islands (theta-resampling)", lambda: (alg.pf.permute_islands(A), alg.pf.sync()))
cand = alg._batch(alg.theta, Nx); cand.step_async(50); cand.sync()
acc = rng.random(Ntheta) < 0.3
timed("accept_islands_from", lambda: (alg.pf.accept_islands_from(cand, acc), alg.pf.sync()))
t0 = time.perf_counter()
alg2 = smc2.SMC2(ssm_cls=mk, prior=prior, data=y, init_Nx=Nx, N=Ntheta, seed=1, nmcmc=2)
alg2.run()
dt = time.perf_counter() - t0
print("whole SMC^2 run: %.3f s, %d resample-move events taking %.3f s" % (dt, len(alg2.move_times), sum(alg2.move_times)))
