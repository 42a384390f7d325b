"""SMC^2 (Chopin, Jacob & Papaspiliopoulos 2013; particles/smc_samplers.py:1038-1167) with the
theta level on the device.

N_theta parameter particles, each carrying a particle filter of N_x state particles: here ALL of
them are islands of ONE device-resident filter, and the outer loop over time costs no host
round trip per step:

* `SMC2.logG` (smc_samplers.py:1099-1120: ``next(pf)`` for every theta, ``lpyt[m] = pf.loglt``) is one
  batched step launch plus `k_theta_update`, which adds the islands' evidence increments to the
  theta log-weights and evaluates the theta-level ESS **on the device**;
* when the ESS drops below ``ESSrmin * N_theta`` (core.py:181-183 for the outer SMC) the kernel
  freezes the batch; the host enqueues ``sync_every`` steps per synchronisation and only deals
  with the resample-move events;
* theta-level resampling of whole filters (`FancyList` deep copies, smc_samplers.py:319-361) is
  `smc_filter_permute_islands`; the PMCMC move (`current_target`, :1129-1143: re-run every
  filter from 0 to t on the proposed theta) is a second batch stepped to t in one call and
  `smc_filter_copy_islands` for the accepted ones;
* the exchange step (:1159-1163, N_x doubled when the acceptance rate is low) is a new batch
  with 2 N_x particles run to t, the theta-weights picking up the ratio of the two evidences.

The MCMC kernel is a Gaussian random walk on theta with covariance (2.38^2 / d) x the weighted
covariance of the theta-particles, ``nmcmc`` sweeps (the reference's default is its waste-free
variant, smc_samplers.py:596-700; the standard resample-move form is what is implemented here).
The prior is any object with ``rvs(size) -> {name: array}`` (or a structured array) and
``logpdf(theta) -> array``; `IndepPrior` covers independent scalar laws.
"""
import ctypes

import numpy as np

from . import _lib
from . import resampling as rs
from . import state_space_models as ssm
from ._lib import check, lib
from .core import SMC


class IndepPrior:
    """Independent scalar priors, host side: ``IndepPrior(sigma=("lognormal", -0.7, 0.5),
    rho=("uniform", 0.0, 1.0))``.  Laws: normal(mu, sd), lognormal(mu, sd), uniform(a, b),
    gamma(shape, rate), beta(a, b)."""

    def __init__(self, **laws):
        self.laws = dict(laws)
        self.names = list(self.laws)

    def rvs(self, size, rng=None):
        rng = np.random if rng is None else rng
        out = {}
        for k, (kind, a, b) in self.laws.items():
            if kind == "normal":
                out[k] = a + b * rng.standard_normal(size)
            elif kind == "lognormal":
                out[k] = np.exp(a + b * rng.standard_normal(size))
            elif kind == "uniform":
                out[k] = a + (b - a) * rng.random_sample(size)
            elif kind == "gamma":
                out[k] = rng.gamma(a, 1.0 / b, size)
            elif kind == "beta":
                out[k] = rng.beta(a, b, size)
            else:
                raise ValueError("unknown law %r" % kind)
        return out

    def logpdf(self, theta):
        """Sum of the scalar log-densities (closed forms in NumPy: importing scipy.stats here would
        cost the first resample-move event a quarter of a second)."""
        from math import lgamma, log, pi
        lp = 0.0
        with np.errstate(invalid="ignore", divide="ignore"):
            for k, (kind, a, b) in self.laws.items():
                v = np.asarray(theta[k], dtype=float)
                if kind == "normal":
                    lp = lp - 0.5 * ((v - a) / b) ** 2 - log(b) - 0.5 * log(2.0 * pi)
                elif kind == "lognormal":
                    lv = np.log(np.where(v > 0, v, 1.0))
                    lp = lp + np.where(v > 0, -0.5 * ((lv - a) / b) ** 2 - log(b) - 0.5 * log(2.0 * pi) - lv, -np.inf)
                elif kind == "uniform":
                    lp = lp + np.where((v >= a) & (v <= b), -log(b - a), -np.inf)
                elif kind == "gamma":             # shape a, rate b
                    lv = np.log(np.where(v > 0, v, 1.0))
                    lp = lp + np.where(v > 0, a * log(b) - lgamma(a) + (a - 1.0) * lv - b * v, -np.inf)
                elif kind == "beta":
                    ok = (v > 0) & (v < 1)
                    vs = np.where(ok, v, 0.5)
                    lp = lp + np.where(ok, lgamma(a + b) - lgamma(a) - lgamma(b) + (a - 1.0) * np.log(vs)
                                       + (b - 1.0) * np.log1p(-vs), -np.inf)
                else:
                    raise ValueError("unknown law %r" % kind)
        return lp


def _as_dict(theta):
    if isinstance(theta, dict):
        return {k: np.asarray(v, dtype=float) for k, v in theta.items()}
    return {k: np.asarray(theta[k], dtype=float) for k in theta.dtype.names}


class SMC2:
    """``SMC2(ssm_cls, prior, data, init_Nx, N)``: N theta-particles x init_Nx state particles.

    ssm_cls : a model class of the fused family (``ssm_cls(**theta)``), e.g. ``kalman.LinearGauss``
    prior   : rvs(size) / logpdf(theta), theta a dict (or structured array) of scalar parameters
    fk_cls  : ``Bootstrap`` (default) or ``GuidedPF``
    ESSrmin : theta-level resampling threshold; nmcmc: random-walk sweeps per move step
    ar_to_increase_Nx : double N_x (exchange step) when the acceptance rate falls below it (< 0: never)
    sync_every : time steps enqueued per host synchronisation
    wastefree, len_chain : the waste-free variant (Dau & Chopin 2022; the reference's default,
        smc_samplers.py:669-684, 730-768): the population is N x len_chain theta-particles; a move
        resamples N of them and runs from each a chain of len_chain - 1 PMMH steps of which EVERY state is
        kept -- with its particle filter.  (``nmcmc`` is then len_chain - 1; ``self.N`` is the population.)
    """

    def __init__(self, ssm_cls=None, prior=None, data=None, init_Nx=100, N=100, fk_cls=None, ESSrmin=0.5,
                 nmcmc=3, ar_to_increase_Nx=-1.0, smc_options=None, seed=None, sync_every=16, max_Nx=1 << 16,
                 wastefree=False, len_chain=10):
        self.ssm_cls, self.prior, self.data = ssm_cls, prior, list(data)
        self.fk_cls = ssm.Bootstrap if fk_cls is None else fk_cls
        self.wastefree = bool(wastefree)
        self.M, self.P = int(N), (int(len_chain) if wastefree else 1)     # chains per move, states kept per chain
        if self.wastefree:
            if self.P < 2:
                raise ValueError("waste-free SMC^2 needs len_chain >= 2")
            N, nmcmc = self.M * self.P, self.P - 1
        self.N, self.Nx, self.ESSrmin, self.nmcmc = int(N), int(init_Nx), ESSrmin, int(nmcmc)
        self.ar_to_increase_Nx, self.max_Nx = ar_to_increase_Nx, max_Nx
        self.smc_options = dict(smc_options or {})
        self.sync_every = int(sync_every)
        self.rng = np.random.RandomState(seed)
        self._seed = int(self.rng.randint(1, 2 ** 31 - 1))
        self.T = len(self.data)
        self.t = 0
        if isinstance(prior, IndepPrior):
            self.theta = _as_dict(prior.rvs(self.N, rng=self.rng))
        else:
            # a prior that draws from numpy's global generator (the reference's StructDist): seeded from
            # this run's own generator for the duration of the call, so that `seed` fixes the run
            np_state = np.random.get_state()
            np.random.seed(self.rng.randint(0, 2 ** 31 - 1))
            try:
                self.theta = _as_dict(prior.rvs(size=self.N))
            finally:
                np.random.set_state(np_state)
        self.names = list(self.theta)
        self.lw = np.zeros(self.N)
        self.logLt = 0.0                 # log evidence of the whole model (outer SMC, core.py:351-359)
        self._finalised = False
        self.ESSs, self.Nxs, self.acc_rates, self.move_times = [], [self.Nx], [], []
        self.acc_fractions = []                    # realised acceptance fractions of the PMCMC steps (acc_rates: probabilities)
        # per step, as the outer particles.SMC would collect them (collectors.py:278-295): the model's
        # log-evidence after step t, and whether a resample-move preceded step t
        self.logLts, self.move_steps = [], []
        self._nbatch = 0
        self._lw_at_reset = np.zeros(self.N)
        self._lo, self._hi = self._my_slice()
        self.pf = self._batch(self.theta, self.Nx, theta_level=True)

    # ------------------------------------------------------------------ batches of filters
    def _my_slice(self):
        """The theta-particles whose filters live in this process (all of them here)."""
        return 0, self.N

    def _batch(self, theta, Nx, theta_level=False, whole=False, span=None):
        """One filter per theta-particle of my slice (``whole``: of ALL the theta given -- the chains of a
        waste-free move; ``span`` = (lo, hi): of that range of them -- a rank's share of the chains), islands of one
        device filter; the Philox streams are keyed by the GLOBAL theta index (island_offset), so a sharded population
        runs the same filters as a single-process one."""
        lo, hi = (0, len(theta[self.names[0]])) if whole else (self._lo, self._hi)
        if span is not None:
            lo, hi = span
        fks = [self.fk_cls(ssm=self.ssm_cls(**{k: float(theta[k][i]) for k in self.names}), data=self.data)
               for i in range(lo, hi)]
        self._nbatch += 1
        pf = SMC(fk=fks, N=Nx, seed=(self._seed + 7919 * self._nbatch) % (2 ** 31 - 1), collect="off",
                 island_offset=lo, **self.smc_options)
        if not pf._fused:
            raise ValueError("SMC2 needs a state-space model of the fused family")
        if theta_level:
            self._enable_theta_level(pf)
        return pf

    # ---- the four places where a sharded population differs (ShardedSMC2 overrides them)
    def _enable_theta_level(self, pf):
        check(lib().smc_filter_theta_enable(pf._f, float(self.ESSrmin)))

    def _evidences(self, pf):
        """log-evidence of the filter of EVERY theta-particle, (N,)."""
        return pf.logLts_islands

    def _resample_filters(self, A):
        check(lib().smc_filter_theta_resume(self.pf._f, None))     # time records back to t first
        self.pf.permute_islands(A)

    def _adopt(self, new, liw):
        """Exchange step: `new` (2 N_x particles, at step t) replaces the batch; theta-weights liw."""
        self._enable_theta_level(new)                                              # from step t on
        check(lib().smc_filter_theta_resume(new._f, liw[self._lo:self._hi].ctypes.data_as(_lib.c_vp)))

    def _theta_state(self, pf):
        lw = np.empty(self.N)
        stop, done = ctypes.c_int64(0), ctypes.c_int64(0)
        ess = np.zeros(self.T)
        check(lib().smc_filter_theta_state(pf._f, lw.ctypes.data_as(_lib.c_vp), ctypes.byref(stop),
                                           ctypes.byref(done), ess.ctypes.data_as(_lib.c_vp)))
        return lw, int(stop.value), int(done.value), ess

    def _theta_logmeans(self, pf):
        """log-mean theta weight after every step the device has accounted for, (T,) (zeros beyond)."""
        out = np.zeros(self.T)
        done = ctypes.c_int64(0)
        check(lib().smc_filter_theta_logmeans(pf._f, out.ctypes.data_as(_lib.c_vp), ctypes.byref(done)))
        return out

    @property
    def rs_flags(self):
        """rs_flags[t]: did a resample-move precede step t (the outer SMC's rs_flag, core.py:326-337)"""
        return [t in self.move_steps for t in range(len(self.ESSs))]

    @property
    def W(self):
        w = np.exp(self.lw - self.lw.max())
        return w / w.sum()

    def _log_mean(self, lw):
        m = lw.max()
        return m + np.log(np.mean(np.exp(lw - m)))

    # ------------------------------------------------------------------ the outer loop
    def _finalise(self):
        """The last evidence term, exactly once (run() on an exhausted sampler is a no-op, as the reference's)."""
        if not self._finalised and self.t >= self.T:
            self.logLt += self._log_mean(self.lw) - self._log_mean(self._lw_at_reset)
            self._finalised = True

    def run(self):
        while self.t < self.T:
            k = min(self.sync_every, self.T - self.t)
            self.pf.step_async(k)
            lw, stop, done, ess = self._theta_state(self.pf)
            t_new = stop if stop else done
            self.ESSs.extend(ess[self.t:t_new].tolist())
            lm = self._theta_logmeans(self.pf)
            self.logLts.extend((self.logLt + lm[self.t:t_new] - self._log_mean(self._lw_at_reset)).tolist())
            # evidence of the whole model: log-mean of the theta weights since the last reset
            self.lw = lw
            self.t = t_new
            self.pf.t = self.pf._n = t_new
            self.pf._invalidate()
            if stop:
                self._resample_move()
        self._finalise()
        return self

    def _resample_move(self):
        import time
        t0 = time.perf_counter()
        self.move_steps.append(self.t)
        # ---- outer evidence up to here, then theta-level resampling (core.py:326-337)
        self.logLt += self._log_mean(self.lw) - self._log_mean(self._lw_at_reset)
        W = self.W
        np_state = np.random.get_state()
        np.random.seed(self.rng.randint(0, 2 ** 31 - 1))
        A = np.asarray(rs.systematic(W, M=self.M if self.wastefree else self.N))
        np.random.set_state(np_state)
        mean = {k: float(np.sum(W * v)) for k, v in self.theta.items()}
        X = np.stack([self.theta[k] for k in self.names], axis=1)
        mu = np.array([mean[k] for k in self.names])
        cov = (X - mu).T @ ((X - mu) * W[:, None])
        d = len(self.names)
        L = np.linalg.cholesky((2.38 ** 2 / d) * cov + 1e-12 * np.eye(d))
        if self.wastefree:
            acc_rate = self._wastefree_move(A, L, d)
            self._maybe_exchange(acc_rate)
            self.Nxs.append(self.Nx)
            self.move_times.append(time.perf_counter() - t0)
            return
        self._resample_filters(A)
        self.theta = {k: v[A].copy() for k, v in self.theta.items()}
        self.lw = np.zeros(self.N)
        self._lw_at_reset = np.zeros(self.N)
        # ---- PMCMC move (smc_samplers.py:1129-1143): candidates re-run from 0 to t
        lp_cur = np.asarray(self.prior.logpdf(self.theta), dtype=float) + self._evidences(self.pf)
        # the rate that drives the exchange step is the reference's: the MEAN ACCEPTANCE PROBABILITY of a step
        # (smc_samplers.py:609 records mean(pb_acc)), averaged over the steps of the move (SMC2.logG)
        pb_rates = []
        for sweep in range(self.nmcmc):
            Z = self.rng.standard_normal((self.N, d)) @ L.T
            prop = {k: self.theta[k] + Z[:, j] for j, k in enumerate(self.names)}
            with np.errstate(all="ignore"):
                lprior = np.asarray(self.prior.logpdf(prop), dtype=float)
            ok = np.isfinite(lprior)
            safe = {k: np.where(ok, prop[k], self.theta[k]) for k in self.names}   # a valid model for every island
            cand = self._batch(safe, self.Nx)
            cand.step_async(self.t)
            lp_prop = np.where(ok, lprior + self._evidences(cand), -np.inf)
            lp_before = lp_cur
            acc = np.log(self.rng.random_sample(self.N)) < lp_prop - lp_cur
            acc &= ok
            self.pf.accept_islands_from(cand, acc[self._lo:self._hi])
            self.theta = {k: np.where(acc, prop[k], self.theta[k]) for k in self.names}
            lp_cur = np.where(acc, lp_prop, lp_cur)
            with np.errstate(all="ignore"):
                pb_rates.append(float(np.mean(np.where(ok, np.exp(np.minimum(lp_prop - lp_before, 0.0)), 0.0))))
            # (the reference stores the mean acceptance PROBABILITY, smc_samplers.py:607-611; the realised
            #  fraction -- a noisier estimate of the same quantity -- under a name of its own)
            self.acc_rates.append(pb_rates[-1])
            self.acc_fractions.append(float(np.mean(acc)))
            del cand
        self._maybe_exchange(float(np.mean(pb_rates)) if pb_rates else 1.0)
        self.Nxs.append(self.Nx)
        self.move_times.append(time.perf_counter() - t0)

    def _wastefree_move(self, A, L, d):
        """MCMCSequenceWF (smc_samplers.py:669-684) on device filters: the M resampled theta-particles start M
        chains of P - 1 PMMH steps (a candidate batch of M filters run from 0 to t per step, as in the standard
        move); every state of every chain stays, together with its filter -- the new population is the
        concatenation [x_0, x_1, .., x_{P-1}] of M theta-particles each, as the reference builds it."""
        M, P, t = self.M, self.P, self.t
        if getattr(self, "device_theta", True):                         # (ShardedSMC2 over the host star / group=None: the
            check(lib().smc_filter_theta_resume(self.pf._f, None))      #  theta level is the host's) time records back to t
        ev_all = self._evidences(self.pf)
        cur = self._batch({k: v[A] for k, v in self.theta.items()}, self.Nx, whole=True)
        cur.take_islands_from(self.pf, A)                               # the resampled filters themselves
        th = {k: v[A].copy() for k, v in self.theta.items()}
        lp = np.asarray(self.prior.logpdf(th), dtype=float) + ev_all[A]
        states, thetas, ars, pbs = [cur], [th], [], []
        for k in range(1, P):
            Z = self.rng.standard_normal((M, d)) @ L.T
            prop = {n_: th[n_] + Z[:, j] for j, n_ in enumerate(self.names)}
            with np.errstate(all="ignore"):
                lprior = np.asarray(self.prior.logpdf(prop), dtype=float)
            ok = np.isfinite(lprior)
            safe = {n_: np.where(ok, prop[n_], th[n_]) for n_ in self.names}
            cand = self._batch(safe, self.Nx, whole=True)
            cand.step_async(t)
            lp_prop = np.where(ok, lprior + cand.logLts_islands, -np.inf)
            lp_old = lp
            acc = (np.log(self.rng.random_sample(M)) < lp_prop - lp) & ok
            nxt = self._batch(th, self.Nx, whole=True)                  # x = x.copy(): the chain's next state
            nxt.take_islands_from(states[-1], np.arange(M))
            nxt.accept_islands_from(cand, acc)
            th = {n_: np.where(acc, prop[n_], th[n_]) for n_ in self.names}
            lp = np.where(acc, lp_prop, lp)
            states.append(nxt)
            thetas.append(th)
            ars.append(float(np.mean(acc)))
            with np.errstate(all="ignore"):
                pbs.append(float(np.mean(np.where(ok, np.exp(np.minimum(lp_prop - lp_old, 0.0)), 0.0))))
            del cand
        self.acc_rates.extend(pbs)                 # (smc_samplers.py:681-682: mean acceptance probabilities)
        self.acc_fractions.extend(ars)
        # ---- the new population: all states of all chains, one batch of M P filters at time t
        self.theta = {n_: np.concatenate([th_[n_] for th_ in thetas]) for n_ in self.names}
        new = self._batch(self.theta, self.Nx)
        for k, st in enumerate(states):
            new.take_islands_from(st, np.arange(M), np.arange(k * M, (k + 1) * M))
        self._enable_theta_level(new)                                   # from step t on, weights zero
        self.pf = new
        self.lw = np.zeros(self.N)
        self._lw_at_reset = np.zeros(self.N)
        return float(np.mean(pbs)) if pbs else 1.0      # mean acceptance PROBABILITY over the steps (smc_samplers.py:609)

    def _maybe_exchange(self, acc_rate):
        # ---- exchange step (smc_samplers.py:1159-1163): more state particles when moves get rejected
        if 0.0 <= acc_rate < self.ar_to_increase_Nx and 2 * self.Nx <= self.max_Nx:
            new = self._batch(self.theta, 2 * self.Nx)
            new.step_async(self.t)
            liw = np.ascontiguousarray(self._evidences(new) - self._evidences(self.pf))
            self._adopt(new, liw)
            self.pf = new
            self.Nx *= 2
            self.lw = liw.copy()
            self._lw_at_reset = np.zeros(self.N)       # E[exp(liw)] = 1 under the extended target

    # ------------------------------------------------------------------ summaries
    def posterior_mean(self):
        W = self.W
        return {k: float(np.sum(W * v)) for k, v in self.theta.items()}

    def posterior_sd(self):
        W, m = self.W, self.posterior_mean()
        return {k: float(np.sqrt(np.sum(W * (v - m[k]) ** 2))) for k, v in self.theta.items()}


class ShardedSMC2(SMC2):
    """SMC^2 with the theta-population sharded over the GPUs of a node (one process per GPU,
    ``group`` = `particles_amd.distributed.Group`): rank r holds the filters of theta-particles
    r M .. r M + M - 1 (M = N / world) as islands of its device filter.

    The theta level itself (N values of theta, N log-weights) is replicated: every rank draws the
    same prior sample, the same resampling uniforms and the same random-walk proposals from one
    seeded host generator, so the only data that crosses GPUs is

    * per time step, the all-gather of the filters' evidence increments (N x 8 bytes over RCCL:
      `Group.gather_evidence`) -- the theta-level ESS needs all of them, this is the exchange step
      of the path;
    * at a theta-resampling, whole filters: `Group.migrate_islands` (packed island states through one
      all-to-all of ncclSend / ncclRecv pairs over xGMI; N_x x (16 d + 40) bytes per island and more
      with history).

    Philox streams are keyed by the global theta index and migration keeps them tied to the slot,
    so the run is the SAME run for any world size (tests: world 2 == world 1, bit for bit)."""

    def __init__(self, group=None, device_theta=None, **kw):
        self.group = group
        world = group.world if group is not None else 1
        # the theta level on the device (smc_filter_theta_enable_sharded: an all-gather of the increments
        # enqueued behind every step, `sync_every` steps per host synchronisation) wherever the group
        # has its device collective; over the host star (CPU tests without RCCL) the per-step form
        has_comm = group is not None and group.comm is not None
        self.device_theta = has_comm if device_theta is None else bool(device_theta)
        if self.device_theta and not has_comm:
            raise ValueError("ShardedSMC2(device_theta=True) needs a Group with its device collective (RCCL)")
        if kw.get("N", 100) % world:
            raise ValueError("ShardedSMC2: N must be a multiple of the number of ranks")
        if not isinstance(kw.get("prior"), IndepPrior):
            raise ValueError("ShardedSMC2 draws the replicated prior sample from its own seeded generator: "
                             "use an IndepPrior")
        if kw.get("seed") is None:
            raise ValueError("ShardedSMC2 needs a seed (the same on every rank)")
        if kw.get("wastefree") and kw.get("N", 100) % world:
            raise ValueError("ShardedSMC2(wastefree=True): the number of chains N must be a multiple of the number of ranks")
        super().__init__(**kw)
        self._cum0 = np.zeros(self.N)        # filters' evidences at the last reset of the theta-weights
        self._lw0 = np.zeros(self.N)         # theta log-weights at that reset

    def _my_slice(self):
        if self.group is None:
            return 0, self.N
        M = self.N // self.group.world
        return self.group.rank * M, (self.group.rank + 1) * M

    def _enable_theta_level(self, pf):
        if self.device_theta:                # replicated theta level fed by ncclAllGather on the stream
            check(lib().smc_filter_theta_enable_sharded(pf._f, self.group.comm, float(self.ESSrmin)))
        # (else: the theta-level ESS needs every rank's increments: done on the host in run())

    def _evidences(self, pf):
        local = pf.logLts_islands
        return local if self.group is None else self.group.gather_evidence(local)

    def _resample_filters(self, A):
        if self.device_theta:
            check(lib().smc_filter_theta_resume(self.pf._f, None))     # time records back to t first
        if self.group is None:
            self.pf.permute_islands(A)
        else:
            self.group.migrate_islands(self.pf, A)

    def _wastefree_move(self, A, L, d):
        """MCMCSequenceWF (smc_samplers.py:669-684) on a population sharded over ranks: rank r runs the chains
        r M/R .. (r + 1) M/R - 1 -- their starting filters arrive from wherever the resampled theta-particles live
        (Group.move_islands), the proposals, acceptance uniforms and theta values are replicated (one seeded host
        generator, the candidates' evidences all-gathered), every state of every chain stays with its filter, and the
        new population [x_0, .., x_{P-1}] is assembled across ranks by P more moves of whole filters.  Same Philox keys
        (global chain / theta index), same batch seeds, same arithmetic: the same run as one process, bit for bit."""
        if self.group is None:
            return super()._wastefree_move(A, L, d)
        M, P, t, grp = self.M, self.P, self.t, self.group
        R, r = grp.world, grp.rank
        Mr = M // R
        clo, chi = r * Mr, (r + 1) * Mr
        if self.device_theta:
            check(lib().smc_filter_theta_resume(self.pf._f, None))      # time records back to t
        ev_all = self._evidences(self.pf)
        th = {k: v[A].copy() for k, v in self.theta.items()}            # (replicated: all M chains)
        cur = self._batch(th, self.Nx, span=(clo, chi))
        grp.move_islands(self.pf, cur, np.arange(M), A)                 # the resampled filters themselves
        lp = np.asarray(self.prior.logpdf(th), dtype=float) + ev_all[A]
        states, thetas, ars, pbs = [cur], [th], [], []
        for k in range(1, P):
            Z = self.rng.standard_normal((M, d)) @ L.T
            prop = {n_: th[n_] + Z[:, j] for j, n_ in enumerate(self.names)}
            with np.errstate(all="ignore"):
                lprior = np.asarray(self.prior.logpdf(prop), dtype=float)
            ok = np.isfinite(lprior)
            safe = {n_: np.where(ok, prop[n_], th[n_]) for n_ in self.names}
            cand = self._batch(safe, self.Nx, span=(clo, chi))
            cand.step_async(t)
            lp_prop = np.where(ok, lprior + grp.gather_evidence(cand.logLts_islands), -np.inf)
            lp_old = lp
            acc = (np.log(self.rng.random_sample(M)) < lp_prop - lp) & ok
            nxt = self._batch(th, self.Nx, span=(clo, chi))             # x = x.copy(): the chain's next state
            nxt.take_islands_from(states[-1], np.arange(Mr))
            nxt.accept_islands_from(cand, acc[clo:chi])
            th = {n_: np.where(acc, prop[n_], th[n_]) for n_ in self.names}
            lp = np.where(acc, lp_prop, lp)
            states.append(nxt)
            thetas.append(th)
            ars.append(float(np.mean(acc)))
            with np.errstate(all="ignore"):
                pbs.append(float(np.mean(np.where(ok, np.exp(np.minimum(lp_prop - lp_old, 0.0)), 0.0))))
            del cand
        self.acc_rates.extend(pbs)
        self.acc_fractions.extend(ars)
        self.theta = {n_: np.concatenate([th_[n_] for th_ in thetas]) for n_ in self.names}
        new = self._batch(self.theta, self.Nx)
        for k, st in enumerate(states):
            grp.move_islands(st, new, k * M + np.arange(M), np.arange(M))
        self._enable_theta_level(new)                                   # from step t on, weights zero
        self.pf = new
        self.lw = np.zeros(self.N)
        self._lw_at_reset = np.zeros(self.N)
        return float(np.mean(pbs)) if pbs else 1.0

    def _adopt(self, new, liw):
        if self.device_theta:                # every rank holds all N theta-weights
            self._enable_theta_level(new)
            check(lib().smc_filter_theta_resume(new._f, np.ascontiguousarray(liw).ctypes.data_as(_lib.c_vp)))

    def _resample_move(self):
        super()._resample_move()
        if not self.device_theta:
            self._cum0 = self._evidences(self.pf)      # after the moves: accepted filters changed theirs
            self._lw0 = self.lw.copy()                 # zeros, or the exchange step's evidence ratios

    def run(self):
        if self.device_theta:
            return super().run()
        while self.t < self.T:
            self.pf.step_async(1)
            self.t += 1
            self.lw = self._lw0 + (self._evidences(self.pf) - self._cum0)
            w = np.exp(self.lw - self.lw.max())
            ess = float(w.sum() ** 2 / np.sum(w * w))
            self.ESSs.append(ess)
            self.logLts.append(self.logLt + self._log_mean(self.lw) - self._log_mean(self._lw_at_reset))
            # (never after the last step: the device theta level -- k_theta_update: t + 1 < T -- and the
            #  reference's outer SMC stop there; a move at t = T would re-run every filter for nothing
            #  and make world-1 runs differ from the one-GPU class)
            if ess < self.ESSrmin * self.N and self.t < self.T:
                self._resample_move()
        self.logLt += self._log_mean(self.lw) - self._log_mean(self._lw_at_reset)
        return self
