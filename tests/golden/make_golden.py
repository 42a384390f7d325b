"""Generate golden vectors by RUNNING THE REFERENCE (nchopin/particles).

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference is imported read-only from /root/reference with the numba stub
in oracle/numba_shim (numba is not installed; the jitted functions then run
as plain Python and compute the same values).  Outputs are the reference's own
results on seeded inputs; they pin the CPU oracle (tests/test_oracle_golden.py)
which in turn checks the HIP path.  /root/reference does not exist on the GPU
box, so nothing else may import it.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle", "numba_shim"), "/root/reference"]

import numpy as np  # noqa: E402
import particles  # noqa: E402
from particles import distributions as dists  # noqa: E402
from particles import kalman  # noqa: E402
from particles import resampling as rs  # noqa: E402
from particles import state_space_models as ssm  # noqa: E402


class ToySSM(ssm.StateSpaceModel):          # README.md:66-72
    def PX0(self):
        return dists.Normal()

    def PX(self, t, xp):
        return dists.Normal(loc=xp)

    def PY(self, t, xp, x):
        return dists.Normal(loc=x, scale=self.sigma)


class Indep2(ssm.StateSpaceModel):
    """A bivariate model assembled with IndepProd (distributions.py:1066-1106): Gaussian
    states, one Gaussian and one Poisson observation."""
    def PX0(self):
        return dists.IndepProd(dists.Normal(scale=1.0), dists.Normal(scale=2.0))

    def PX(self, t, xp):
        return dists.IndepProd(dists.Normal(loc=0.9 * xp[:, 0]),
                               dists.Normal(loc=0.5 * xp[:, 1] + 0.1 * xp[:, 0], scale=0.7))

    def PY(self, t, xp, x):
        return dists.IndepProd(dists.Normal(loc=x[:, 0], scale=0.5),
                               dists.Poisson(rate=np.exp(0.3 * x[:, 1])))


def run_case(model, fk_cls, T, N, scheme, ESSrmin, data_seed=42, run_seed=123):
    np.random.seed(data_seed)
    x, y = model.simulate(T)
    np.random.seed(run_seed)
    pf = particles.SMC(fk=fk_cls(ssm=model, data=y), N=N, resampling=scheme,
                       ESSrmin=ESSrmin)
    pf.run()
    return dict(
        y=np.array(y), T=T, N=N, scheme=scheme, ESSrmin=ESSrmin,
        data_seed=data_seed, run_seed=run_seed,
        ESSs=np.array(pf.summaries.ESSs), logLts=np.array(pf.summaries.logLts),
        rs_flags=np.array(pf.summaries.rs_flags), logLt=pf.logLt,
        X=pf.X, A=pf.A, lw=pf.wgts.lw, W=pf.W)


def run_sqmc_case(model, fk_cls, T, N, data_seed=42, qmc_seed=7):
    """SMC(qmc=True) (core.py:315-349).  rqmc.sobol draws from scipy's self-seeded engine, so
    the run is made repeatable -- and its points recorded -- by giving the engine a seed."""
    from scipy.stats import qmc
    from particles import rqmc
    np.random.seed(data_seed)
    x, y = model.simulate(T)
    tape = []
    ss = np.random.SeedSequence(qmc_seed)

    def seeded_sobol(N_, d):
        eng = qmc.Sobol(d, seed=np.random.default_rng(ss.spawn(1)[0]))
        u = eng.random(N_)
        v = 0.5 + (1.0 - rqmc.TOL) * (u - 0.5)            # rqmc.py:9-13 safe_generate
        tape.append(v)
        return v

    orig = rqmc.sobol
    rqmc.sobol = seeded_sobol
    try:
        pf = particles.SMC(fk=fk_cls(ssm=model, data=y), N=N, qmc=True)
        pf.run()
    finally:
        rqmc.sobol = orig
    d = dict(y=np.array(y), T=T, N=N, scheme="systematic", ESSrmin=0.5, data_seed=data_seed,
             run_seed=qmc_seed, ESSs=np.array(pf.summaries.ESSs),
             logLts=np.array(pf.summaries.logLts), rs_flags=np.array(pf.summaries.rs_flags),
             logLt=pf.logLt, X=pf.X, A=pf.A, lw=pf.wgts.lw, W=pf.W, u0=tape[0],
             u=np.array(tape[1:]))
    return d


def main():
    out = {}

    # --- SMC runs (C1 and reduced-size C3/C4 of BASELINE.json) -------------
    for scheme in ("systematic", "stratified", "multinomial"):
        out["toy_%s" % scheme] = run_case(ToySSM(sigma=0.2), ssm.Bootstrap,
                                          200, 1000, scheme, 0.5)
        out["sv_%s" % scheme] = run_case(ssm.StochVol(), ssm.Bootstrap,
                                         50, 2048, scheme, 1.0)
    # adaptive resampling (some steps do not resample), odd N
    out["lg_adaptive"] = run_case(kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=1.5),
                                  ssm.Bootstrap, 100, 777, "systematic", 0.5)
    out["lg_guided"] = run_case(kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=0.2),
                                ssm.GuidedPF, 60, 500, "systematic", 0.5)
    for dx in (4, 32):
        mv = kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=dx)
        out["mv%d_guided" % dx] = run_case(mv, ssm.GuidedPF, 12, 256, "systematic", 0.5)
        out["mv%d_boot" % dx] = run_case(mv, ssm.Bootstrap, 12, 256, "stratified", 0.5)

    # --- further univariate models of the fused family (state_space_models.py:546-577, 657-683)
    out["gordon_boot"] = run_case(ssm.Gordon_etal(), ssm.Bootstrap, 40, 600, "systematic", 0.5)
    out["theta_boot"] = run_case(ssm.ThetaLogistic(), ssm.Bootstrap, 40, 600, "stratified", 0.5)
    out["svlev_boot"] = run_case(ssm.StochVolLeverage(phi=-0.5), ssm.Bootstrap, 40, 600,
                                 "systematic", 0.5)
    # Poisson observations (state_space_models.py:611-630, distributions.py:519-532)
    out["cox_boot"] = run_case(ssm.DiscreteCox(mu=0.5, sigma=0.4, phi=0.9), ssm.Bootstrap, 40, 600,
                               "systematic", 0.5)

    out["indep_boot"] = run_case(Indep2(), ssm.Bootstrap, 30, 400, "systematic", 0.5)

    # --- SQMC (core.py:315-349): quasi-random points, Hilbert (sorted) order, ppf moves
    out["sqmc_toy"] = run_sqmc_case(kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=0.5),
                                    ssm.Bootstrap, 30, 512)
    out["sqmc_sv"] = run_sqmc_case(ssm.StochVol(), ssm.Bootstrap, 25, 300)
    out["sqmc_guided"] = run_sqmc_case(kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=0.3),
                                       ssm.GuidedPF, 25, 256)

    out["sqmc_mv2"] = run_sqmc_case(kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=2),
                                    ssm.Bootstrap, 12, 256)
    out["sqmc_mv3_guided"] = run_sqmc_case(kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=3),
                                           ssm.GuidedPF, 8, 128)
    # --- the Hilbert codec and sort on their own (hilbert.py:13-58)
    from particles import hilbert
    rng = np.random.default_rng(17)
    hb = {}
    for d in (2, 3, 5, 8):
        maxint = np.floor(2 ** (62 / d))
        xint = np.floor(rng.random((200, d)) * maxint).astype(np.int64)
        xint[:6] = [[0] * d, [1] + [0] * (d - 1), [0] * (d - 1) + [1], [int(maxint) - 1] * d,
                    [3] * d, [2 ** 10] + [5] * (d - 1)]
        x = rng.standard_normal((300, d)) * (1.0 + np.arange(d))
        hb["xint%d" % d] = xint
        hb["h%d" % d] = hilbert.hilbert_array(xint)
        hb["x%d" % d] = x
        hb["order%d" % d] = hilbert.hilbert_sort(x)
    out["hilbert"] = hb

    # --- auxiliary particle filter (core.py:299-313), Pitt & Shephard's StochVol proposal
    out["sv_apf"] = run_case(ssm.StochVol(), ssm.AuxiliaryPF, 30, 500, "systematic", 0.5)
    out["lg_apf"] = run_case(kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=0.6), ssm.AuxiliaryPF, 30, 500,
                             "systematic", 0.5)
    out["sv_guided"] = run_case(ssm.StochVol(), ssm.GuidedPF, 30, 500, "systematic", 0.5)
    # AuxiliaryBootstrap (state_space_models.py:431-438): the bootstrap move with the auxiliary weights
    out["sv_apfboot"] = run_case(ssm.StochVol(), ssm.AuxiliaryBootstrap, 30, 500, "systematic", 0.5)
    out["lg_apfboot"] = run_case(kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=0.6), ssm.AuxiliaryBootstrap, 30, 500,
                                 "stratified", 0.6)
    # AuxiliaryPF of the multivariate model (kalman.py:348-361: optimal proposal, logeta = log p(y_{t+1} | x_t))
    out["mv_apf"] = run_case(kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=4), ssm.AuxiliaryPF, 16, 400,
                             "systematic", 0.7)

    # --- full particle history + genealogy (smoothing.py:181-255), adaptive
    # resampling so that some A_t are arange ---------------------------------
    np.random.seed(42)
    model = kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=1.5)
    x, y = model.simulate(30)
    np.random.seed(123)
    pf = particles.SMC(fk=ssm.Bootstrap(ssm=model, data=y), N=300, resampling="systematic",
                       ESSrmin=0.5, store_history=True)
    pf.run()
    out["history"] = dict(
        y=np.array(y), T=30, N=300, scheme="systematic", ESSrmin=0.5, data_seed=42, run_seed=123,
        logLt=pf.logLt, rs_flags=np.array(pf.summaries.rs_flags),
        hist_X=np.array(pf.hist.X), hist_A=np.array(pf.hist.A[1:]),
        hist_lw=np.array([w.lw for w in pf.hist.wgts]), hist_W=np.array([w.W for w in pf.hist.wgts]),
        A0_is_none=pf.hist.A[0] is None,
        trajectories=np.array(pf.hist.compute_trajectories()))

    # --- Kalman exact log-likelihoods (analytic KAT) ------------------------
    for name, model in (("toy", kalman.LinearGauss(rho=1.0, sigmaX=1.0, sigmaY=0.2, sigma0=1.0)),
                        ("mv32", kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=32))):
        np.random.seed(42)
        T = 200 if name == "toy" else 12
        x, y = (ToySSM(sigma=0.2) if name == "toy" else model).simulate(T)
        kf = kalman.Kalman(ssm=model, data=y)
        kf.filter()
        out["kalman_%s" % name] = dict(
            y=np.array(y), loglik=float(np.sum(kf.logpyt)),
            filt_means=np.array([np.squeeze(f.mean) for f in kf.filt]))

    # --- resampling schemes on fixed weights --------------------------------
    np.random.seed(7)
    lw = 3.0 * np.random.randn(1500)
    lw[[3, 500]] = -np.inf
    W = rs.exp_and_normalise(lw)
    res = dict(W=W, lw=lw)
    for scheme in ("systematic", "stratified", "multinomial"):
        for M in (1500, 400, 4000):
            np.random.seed(11)
            res["A_%s_%d" % (scheme, M)] = rs.resampling(scheme, W, M=M)
    np.random.seed(5)
    res["spacings_100"] = rs.uniform_spacings(100)
    out["resampling"] = res

    # --- residual and killing (resampling.py:611-626, 680-697), same weights --------------
    res2 = dict(W=W)
    for M in (1500, 400, 4000):
        np.random.seed(11)
        res2["A_residual_%d" % M] = rs.resampling("residual", W, M=M)
    np.random.seed(11)
    res2["A_killing_1500"] = rs.resampling("killing", W, M=1500)
    for M in (1500, 400, 4000):
        np.random.seed(11)
        res2["A_ssp_%d" % M] = rs.resampling("ssp", W, M=M)
    Wd = np.zeros(64)
    Wd[[3, 17, 40]] = [0.5, 0.25, 0.25]         # M W integral: no residual draw at all
    np.random.seed(11)
    res2["W_integral"] = Wd
    res2["A_residual_integral"] = rs.resampling("residual", Wd, M=64)
    out["resampling2"] = res2

    # --- Weights / log-sum-exp helpers --------------------------------------
    np.random.seed(3)
    lw = 10.0 * np.random.randn(1000)
    lw[10] = np.nan
    lw[20] = -np.inf
    lwc = lw.copy()
    w = rs.Weights(lw=lwc)
    w2 = w.add(np.random.randn(1000))
    out["weights"] = dict(
        lw_in=lw, lw_after=w.lw, W=w.W, ESS=w.ESS, log_mean=w.log_mean,
        W2=w2.W, ESS2=w2.ESS, log_mean2=w2.log_mean, lw2=w2.lw,
        lse=rs.log_sum_exp(w.lw), lme=rs.log_mean_exp(w.lw),
        lme_w=rs.log_mean_exp(w2.lw, W=w.W), essl=rs.essl(w.lw),
        ean=rs.exp_and_normalise(w.lw),
        wq=np.array(rs.wquantiles(w.W, np.sin(np.arange(1000.0)), alphas=(0.05, 0.25, 0.5, 0.75, 0.999)))
        , wq2=rs.wquantiles(w.W, np.stack([np.sin(np.arange(1000.0)), np.cos(3.0 * np.arange(1000.0))], axis=1)),
        wmean=rs.wmean_and_var(w.W, np.sin(np.arange(1000.0)))["mean"],
        wvar=rs.wmean_and_var(w.W, np.sin(np.arange(1000.0)))["var"])

    # --- distributions -------------------------------------------------------
    np.random.seed(9)
    xs = np.random.randn(64)
    loc = np.random.randn(64)
    dd = dict(x=xs, loc=loc,
              normal_logpdf=dists.Normal(loc=loc, scale=0.7).logpdf(xs),
              normal_logpdf_sv=dists.Normal(loc=0.0, scale=np.exp(0.5 * xs)).logpdf(np.array([0.3])))
    np.random.seed(10)
    dd["normal_rvs"] = dists.Normal(loc=loc, scale=0.7).rvs(size=64)
    d = 5
    Amat = np.random.randn(d, d)
    cov = Amat @ Amat.T + d * np.eye(d)
    mloc = np.random.randn(40, d)
    mv = dists.MvNormal(loc=mloc, scale=1.3, cov=cov)
    xs5 = np.random.randn(40, d)
    dd.update(cov=cov, mloc=mloc, x5=xs5, mv_logpdf=mv.logpdf(xs5))
    np.random.seed(12)
    dd["mv_rvs"] = mv.rvs(size=40)
    out["dists"] = dd

    # --- SMC^2 (smc_samplers.py:1038-1167): R independent runs of the REFERENCE's algorithm on one data
    # set -- standard (not waste-free) resample-move with len_chain - 1 random-walk steps, the form the
    # device implements -- recorded run by run: the device's runs must be draws from the same
    # distributions (tests compare means within 3 standard errors)
    if "smc2_ref" in sys.argv[1:] or not sys.argv[1:]:
        from particles import smc_samplers as ssp
        R, T, N, Nx, len_chain = 24, 40, 64, 64, 4
        np.random.seed(4)
        x, y = kalman.LinearGauss(sigmaX=1.0, sigmaY=0.4, rho=0.8).simulate(T)
        prior = dists.StructDist({"rho": dists.Uniform(a=0.3, b=0.99), "sigmaY": dists.Gamma(a=2.0, b=4.0)})
        rec = dict(logLt=[], m_rho=[], m_sigmaY=[], s_rho=[], s_sigmaY=[], ESSs=[], rs_flags=[], Nx_final=[])
        for r in range(R):
            np.random.seed(1000 + r)
            fk = ssp.SMC2(ssm_cls=kalman.LinearGauss, prior=prior, data=y, init_Nx=Nx, len_chain=len_chain,
                          wastefree=False, ar_to_increase_Nx=-1.0)
            alg = particles.SMC(fk=fk, N=N, ESSrmin=0.5)
            alg.run()
            W = alg.W
            for k in ("rho", "sigmaY"):
                m = np.sum(W * alg.X.theta[k])
                rec["m_" + k].append(m)
                rec["s_" + k].append(np.sqrt(np.sum(W * (alg.X.theta[k] - m) ** 2)))
            rec["logLt"].append(alg.logLt)
            rec["ESSs"].append(alg.summaries.ESSs)
            rec["rs_flags"].append(alg.summaries.rs_flags)
            rec["Nx_final"].append(alg.X.pfs[0].N)
            print("smc2_ref run", r, alg.logLt, rec["m_rho"][-1], rec["m_sigmaY"][-1], sum(alg.summaries.rs_flags), flush=True)
        kal = kalman.Kalman(ssm=kalman.LinearGauss(sigmaX=1.0, sigmaY=0.4, rho=0.8), data=y)
        kal.filter()
        out["smc2_ref"] = dict(y=np.array(y), R=R, T=T, N=N, Nx=Nx, len_chain=len_chain, ESSrmin=0.5,
                               prior_rho=np.array([0.3, 0.99]), prior_sigmaY=np.array([2.0, 4.0]),
                               sigmaX=1.0, true_rho=0.8, true_sigmaY=0.4, kalman_loglik_at_truth=np.sum(kal.logpyt),
                               **{k: np.array(v, dtype=float) for k, v in rec.items()})

    # --- the same with the reference's DEFAULT move, waste-free (smc_samplers.py:669-684, 730-768): N chains of
    # len_chain states, all kept -- a population of N x len_chain theta-particles
    if "smc2_wf_ref" in sys.argv[1:] or not sys.argv[1:]:
        from particles import smc_samplers as ssp
        R, T, N, Nx, len_chain = 32, 40, 32, 64, 4
        np.random.seed(4)
        x, y = kalman.LinearGauss(sigmaX=1.0, sigmaY=0.4, rho=0.8).simulate(T)
        prior = dists.StructDist({"rho": dists.Uniform(a=0.3, b=0.99), "sigmaY": dists.Gamma(a=2.0, b=4.0)})
        rec = dict(logLt=[], m_rho=[], m_sigmaY=[], s_rho=[], s_sigmaY=[], ESSs=[], rs_flags=[], Nx_final=[])
        for r in range(R):
            np.random.seed(2000 + r)
            fk = ssp.SMC2(ssm_cls=kalman.LinearGauss, prior=prior, data=y, init_Nx=Nx, len_chain=len_chain,
                          wastefree=True, ar_to_increase_Nx=-1.0)
            alg = particles.SMC(fk=fk, N=N, ESSrmin=0.5)
            alg.run()
            W = alg.W
            assert alg.X.N == N * len_chain
            for k in ("rho", "sigmaY"):
                m = np.sum(W * alg.X.theta[k])
                rec["m_" + k].append(m)
                rec["s_" + k].append(np.sqrt(np.sum(W * (alg.X.theta[k] - m) ** 2)))
            rec["logLt"].append(alg.logLt)
            rec["ESSs"].append(alg.summaries.ESSs)
            rec["rs_flags"].append(alg.summaries.rs_flags)
            rec["Nx_final"].append(alg.X.pfs[0].N)
            print("smc2_wf_ref run", r, alg.logLt, rec["m_rho"][-1], rec["m_sigmaY"][-1], sum(alg.summaries.rs_flags), flush=True)
        out["smc2_wf_ref"] = dict(y=np.array(y), R=R, T=T, N=N, Nx=Nx, len_chain=len_chain, ESSrmin=0.5,
                                  prior_rho=np.array([0.3, 0.99]), prior_sigmaY=np.array([2.0, 4.0]), sigmaX=1.0,
                                  **{k: np.array(v, dtype=float) for k, v in rec.items()})

    # --- the two models without a fused descriptor (state_space_models.py:580-606, 630-655): Bootstrap filters
    out["bearings_boot"] = run_case(ssm.BearingsOnly(), ssm.Bootstrap, 25, 400, "systematic", 0.5)
    rngm = np.random.RandomState(5)
    Fm = 0.8 * np.eye(3) + 0.05 * rngm.randn(3, 3)
    cx = 0.2 * (np.eye(3) + 0.3 * np.ones((3, 3)))
    cy = np.eye(3) + 0.4 * (np.ones((3, 3)) - np.eye(3))
    mvsv = ssm.MVStochVol(mu=np.array([-1.0, -0.5, 0.2]), covX=cx, corY=cy, F=Fm)
    case = run_case(mvsv, ssm.Bootstrap, 25, 400, "stratified", 0.5)
    case.update(F=Fm, covX=cx, corY=cy, mu=np.array([-1.0, -0.5, 0.2]))
    out["mvsv_boot"] = case

    # --- wmean_and_cov, the structured-array moments / quantiles (resampling.py:341-380, 420-442) ----------
    rng = np.random.RandomState(21)
    Wm = rs.exp_and_normalise(2.0 * rng.randn(1500))
    Xm = rng.randn(1500, 5) @ rng.randn(5, 5) + np.arange(5.0)
    m5, c5 = rs.wmean_and_cov(Wm, Xm)
    m1, c1 = rs.wmean_and_cov(Wm, Xm[:, 0])
    xs = np.zeros(1500, dtype=[("a", float), ("b", float)])
    xs["a"], xs["b"] = Xm[:, 1], np.exp(0.3 * Xm[:, 2])
    mv = rs.wmean_and_var_str_array(Wm, xs)
    wq = rs.wquantiles_str_array(Wm, xs, alphas=(0.1, 0.5, 0.9))
    out["moments_cov"] = dict(W=Wm, X=Xm, mean5=m5, cov5=c5, mean1=m1, cov1=np.array(c1), sa=xs["a"], sb=xs["b"],
                              sm_a=mv["mean"]["a"], sm_b=mv["mean"]["b"], sv_a=mv["var"]["a"], sv_b=mv["var"]["b"],
                              sq_a=np.array(wq["a"]), sq_b=np.array(wq["b"]))

    only = sys.argv[1:]              # optional: names of the fixtures to (re)write
    for name, case in out.items():
        if only and name not in only:
            continue
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **case)
        print(name, {k: (v.shape if hasattr(v, "shape") and v.shape else v)
                     for k, v in case.items() if k in ("logLt", "loglik", "X")})


if __name__ == "__main__":
    main()
