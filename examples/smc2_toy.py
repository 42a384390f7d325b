"""SMC^2 in miniature on the island primitives (the algorithm of smc_samplers.py:1038-1167,
Chopin, Jacob & Papaspiliopoulos 2013): N_theta parameter particles, each with its own particle
filter of N_x particles -- all of them islands of ONE device-resident filter.

  * per time step: every filter advances one step (one launch for all of them); the
    theta-weights pick up the incremental evidence log p(y_t | y_{0:t-1}, theta);
  * when the theta-ESS drops: theta-level resampling of whole filters (permute_islands) and a
    PMMH move -- a second batch runs the proposed thetas from 0 to t, accepted ones are taken
    over (accept_islands_from).

Model: X_0 ~ N(0,1), X_t ~ N(X_{t-1}, 1), Y_t ~ N(X_t, sigma^2), sigma unknown, prior
log sigma ~ N(log 0.5, 0.5^2).   Run on a GPU box:  python examples/smc2_toy.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import particles_amd as pa
from particles_amd import kalman, state_space_models as ssm


def log_prior(ls):
    return -0.5 * ((ls - np.log(0.5)) / 0.5) ** 2


def batch(log_sigmas, y, Nx, seed):
    fks = [ssm.Bootstrap(ssm=kalman.ToySSM(float(np.exp(ls))), data=y) for ls in log_sigmas]
    return pa.SMC(fk=fks, N=Nx, seed=seed, collect="off")


def main(T=100, Ntheta=512, Nx=512, sigma_true=0.3, seed=1):
    rng = np.random.default_rng(seed)
    x = np.cumsum(rng.standard_normal(T))
    y = [np.array([v]) for v in x + sigma_true * rng.standard_normal(T)]
    ls = np.log(0.5) + 0.5 * rng.standard_normal(Ntheta)            # theta-particles: log sigma
    lw = np.zeros(Ntheta)
    pf = batch(ls, y, Nx, seed=10)
    nmoves = 0
    t0 = time.perf_counter()
    for t in range(T):
        before = pf.logLts_islands if t else np.zeros(Ntheta)
        pf.step_async(1)
        lw += pf.logLts_islands - before                            # incremental evidence per theta
        W = np.exp(lw - lw.max())
        W /= W.sum()
        if 1.0 / np.sum(W ** 2) < 0.5 * Ntheta:                     # resample-move at the theta level
            A = rng.choice(Ntheta, size=Ntheta, p=W)
            pf.permute_islands(A)
            ls, lw = ls[A], np.zeros(Ntheta)
            step = 2.38 * np.sqrt(np.cov(ls, aweights=None)) if Ntheta > 1 else 0.1
            prop = ls + step * rng.standard_normal(Ntheta)
            cand = batch(prop, y, Nx, seed=1000 + t)
            cand.step_async(t + 1)
            log_ratio = (cand.logLts_islands + log_prior(prop)) - (pf.logLts_islands + log_prior(ls))
            acc = np.log(rng.random(Ntheta)) < log_ratio
            pf.accept_islands_from(cand, acc)
            ls = np.where(acc, prop, ls)
            nmoves += 1
    dt = time.perf_counter() - t0
    W = np.exp(lw - lw.max())
    W /= W.sum()
    mean = float(np.sum(W * np.exp(ls)))
    sd = float(np.sqrt(np.sum(W * (np.exp(ls) - mean) ** 2)))
    print("SMC^2: N_theta=%d x N_x=%d, T=%d, %d resample-move steps, %.2f s" % (Ntheta, Nx, T, nmoves, dt))
    print("posterior of sigma: mean %.3f, sd %.3f   (data simulated with sigma = %.2f)" % (mean, sd, sigma_true))
    return mean, sd


if __name__ == "__main__":
    main()
