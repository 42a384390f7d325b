"""SMC^2 (smc_samplers.py:1038-1167, Chopin, Jacob & Papaspiliopoulos 2013) with the theta level
on the device: N_theta parameter particles, each with its own particle filter of N_x particles --
all of them islands of ONE device-resident filter (particles_amd.smc2).

  * per time step: every filter advances one step and a one-workgroup kernel adds the evidence
    increments log p(y_t | y_{0:t-1}, theta) to the theta weights and checks the theta-level ESS
    -- no host round trip; the host enqueues `sync_every` steps at a time;
  * when the theta-ESS drops the batch freezes itself; the host resamples whole filters
    (permute_islands), runs the PMCMC move on a second batch stepped from 0 to t in one call and
    takes over the accepted ones (accept_islands_from).

Model: X_0 ~ N(0,1), X_t ~ N(X_{t-1}, 1), Y_t ~ N(X_t, sigma^2), sigma unknown, prior
log sigma ~ N(log 0.5, 0.5^2).   Run on a GPU box:  python examples/smc2_toy.py
On several GPUs (theta-population sharded, one process per GPU, smc2.ShardedSMC2):
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/smc2_toy.py
(any launcher that sets RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT will do; torch is not imported).
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from particles_amd import kalman, smc2


def main(T=100, Ntheta=512, Nx=512, sigma_true=0.3, seed=1, sync_every=16):
    rng = np.random.default_rng(seed)
    x = np.cumsum(rng.standard_normal(T))
    y = [np.array([v]) for v in x + sigma_true * rng.standard_normal(T)]
    prior = smc2.IndepPrior(sigmaY=("lognormal", np.log(0.5), 0.5))
    kw = dict(ssm_cls=lambda sigmaY: kalman.LinearGauss(rho=1.0, sigmaX=1.0, sigmaY=sigmaY, sigma0=1.0),
              prior=prior, data=y, init_Nx=Nx, N=Ntheta, seed=seed, nmcmc=2)
    grp = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from particles_amd.distributed import Group
        grp = Group()
        alg = smc2.ShardedSMC2(group=grp, **kw)
    else:
        alg = smc2.SMC2(sync_every=sync_every, **kw)
    t0 = time.perf_counter()
    alg.run()
    dt = time.perf_counter() - t0
    if grp is not None:
        rank = grp.rank
        grp.close()
        if rank:
            return alg.posterior_mean()["sigmaY"], alg.posterior_sd()["sigmaY"]
    mean, sd = alg.posterior_mean()["sigmaY"], alg.posterior_sd()["sigmaY"]
    print("SMC^2: N_theta=%d x N_x=%d, T=%d, %d resample-move steps (%.2f s of the %.2f s), log evidence %.3f"
          % (Ntheta, Nx, T, len(alg.move_times), sum(alg.move_times), dt, alg.logLt))
    print("posterior of sigma: mean %.3f, sd %.3f   (data simulated with sigma = %.2f)" % (mean, sd, sigma_true))
    return mean, sd


if __name__ == "__main__":
    main()
