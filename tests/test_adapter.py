"""particles_amd.adapter against the REAL reference package (build container only: needs
/root/reference; the kernels run through the emulator there).  What INTEGRATION.md section 3
promises: reference objects in, device filter out -- PMMH(smc_cls=HipSMC), particles.SMC swapped
for SMC2-style callers, the resampling registry."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "particles")),
                                reason="the reference package is not on this box")


@pytest.fixture(scope="module")
def ref():
    sys.dont_write_bytecode = True
    for p in (REF, os.path.join(ROOT, "oracle", "numba_shim")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import particles
    from particles import distributions, kalman, mcmc, resampling, state_space_models
    return dict(particles=particles, dists=distributions, kalman=kalman, mcmc=mcmc, rs=resampling,
                ssm=state_space_models)


def test_adapt_maps_stock_models_only(ref):
    import particles_amd as pa
    from particles_amd import adapter
    rk, rssm = ref["kalman"], ref["ssm"]
    y = [np.array([0.1 * t]) for t in range(6)]
    fk = adapter.adapt(rssm.Bootstrap(ssm=rk.LinearGauss(sigmaX=0.7, sigmaY=0.3, rho=0.8), data=y))
    assert isinstance(fk, pa.state_space_models.Bootstrap) and fk.ssm.sigmaX == 0.7 and fk.ssm.rho == 0.8
    assert abs(fk.ssm.sigma0 - 0.7 / np.sqrt(1 - 0.64)) < 1e-15 and fk._device_model() is not None
    g = adapter.adapt(rssm.GuidedPF(ssm=rk.MVLinearGauss_Guarniero_etal(alpha=0.3, dx=3), data=[np.zeros((1, 3))]))
    assert isinstance(g, pa.state_space_models.GuidedPF) and np.allclose(g.ssm.F, rk.MVLinearGauss_Guarniero_etal(alpha=0.3, dx=3).F)
    sv = adapter.adapt(rssm.Bootstrap(ssm=rssm.StochVol(mu=-1.0, rho=0.9, sigma=0.2), data=y))
    assert sv.ssm.mu == -1.0 and sv.ssm.sigma == 0.2

    class MySV(rssm.StochVol):                       # a user's subclass: not ours to reinterpret
        def PY(self, t, xp, x):
            return ref["dists"].Normal(scale=2.0 * np.exp(x / 2.0))
    assert adapter.adapt(rssm.Bootstrap(ssm=MySV(), data=y)) is None


def test_hipsmc_is_a_drop_in_smc_cls_for_pmmh(ref):
    """The reference's PMMH, unchanged, with smc_cls=HipSMC: every likelihood evaluation is a
    device filter (fused loop), the chain is a valid PMMH chain on the same posterior."""
    import particles_amd as pa
    from particles_amd import adapter
    particles, rk, rssm, mcmc, dists = (ref[k] for k in ("particles", "kalman", "ssm", "mcmc", "dists"))
    np.random.seed(3)
    model = rk.LinearGauss(sigmaX=1.0, sigmaY=0.5, rho=0.9)
    x, y = model.simulate(40)
    HipSMC = adapter.HipSMC()
    assert issubclass(HipSMC, particles.SMC)
    made = []
    real_init = pa.SMC.__init__

    def spy(self, *a, **kw):
        real_init(self, *a, **kw)
        made.append(self._fused)
    pa.SMC.__init__ = spy
    try:
        prior = dists.StructDist({"rho": dists.Uniform(a=0.5, b=0.99), "sigmaY": dists.Gamma(a=2.0, b=4.0)})
        pmmh = mcmc.PMMH(niter=12, ssm_cls=rk.LinearGauss, smc_cls=HipSMC, prior=prior, data=y, Nx=2048,
                         theta0=None, adaptive=True)
        pmmh.run()
    finally:
        pa.SMC.__init__ = real_init
    assert len(made) >= 3 and all(made)        # a fused device filter per proposal inside the prior
    lp = pmmh.chain.lpost
    assert lp.shape == (12,) and np.all(np.isfinite(lp))
    # the device's evidence estimate at theta0 agrees with the reference's own filter
    th = {k: float(pmmh.chain.theta[k][0]) for k in ("rho", "sigmaY")}
    lls = []
    for cls in (particles.SMC, HipSMC):
        np.random.seed(5)
        pf = cls(fk=rssm.Bootstrap(ssm=rk.LinearGauss(**th), data=y), N=20000, collect="off")
        pf.run()
        lls.append(pf.logLt)
    assert abs(lls[0] - lls[1]) < 0.2, lls
    # a model outside the table keeps the reference's own path, as a real particles.SMC
    class Toy(rssm.StateSpaceModel):
        def PX0(self): return dists.Normal()
        def PX(self, t, xp): return dists.Normal(loc=xp)
        def PY(self, t, xp, x): return dists.Normal(loc=x, scale=0.3)
    pf = HipSMC(fk=rssm.Bootstrap(ssm=Toy(), data=y), N=100)
    assert isinstance(pf, particles.SMC) and type(pf) is HipSMC
    pf.run()
    assert np.isfinite(pf.logLt)


def test_install_swaps_particles_smc_and_registry(ref):
    from particles_amd import adapter
    import particles_amd as pa
    particles, rk, rssm, rs = ref["particles"], ref["kalman"], ref["ssm"], ref["rs"]
    orig = particles.SMC
    adapter.install()
    try:
        y = [np.array([0.2 * t]) for t in range(10)]
        pf = particles.SMC(fk=rssm.Bootstrap(ssm=rk.LinearGauss(), data=y), N=512, collect="off")
        assert isinstance(pf, pa.SMC)
        pf.run()
        assert np.isfinite(pf.logLt)
    finally:
        adapter.uninstall()
    assert particles.SMC is orig
    names = adapter.register_into(rs)
    assert "systematic_hip" in names and "systematic_hip" in rs.rs_funcs
    W = np.random.default_rng(0).random(3000)
    W /= W.sum()
    for scheme in ("systematic", "stratified", "multinomial"):
        np.random.seed(11)
        A_ref = rs.resampling(scheme, W, M=2500)
        np.random.seed(11)
        A_dev = rs.resampling(scheme + "_hip", W, M=2500)
        assert A_dev.shape == A_ref.shape and np.mean(A_dev == A_ref) > 0.999
    # and inside the reference's own SMC
    np.random.seed(2)
    y = [np.array([0.1 * t]) for t in range(15)]

    class Toy(rssm.StateSpaceModel):
        def PX0(self): return ref["dists"].Normal()
        def PX(self, t, xp): return ref["dists"].Normal(loc=xp)
        def PY(self, t, xp, x): return ref["dists"].Normal(loc=x, scale=0.3)
    lls = []
    for scheme in ("systematic", "systematic_hip"):
        np.random.seed(9)
        pf = particles.SMC(fk=rssm.Bootstrap(ssm=Toy(), data=y), N=400, resampling=scheme)
        pf.run()
        lls.append(pf.logLt)
    assert abs(lls[0] - lls[1]) < 1e-9 * abs(lls[0]), lls
    for n in names:
        del rs.rs_funcs[n]


def test_reference_apf_runs_fused_through_the_adapter(ref):
    """particles.AuxiliaryPF of the stock StochVol handed to HipSMC: N <= 1024 lands on the one-launch
    filter, larger N on the two-level step (both fused);
    the evidence agrees with the reference's own run (same model, same data, its NumPy generator)."""
    import particles_amd as pa
    from particles_amd import adapter
    from parity_cases import describe
    particles, rssm = ref["particles"], ref["ssm"]
    rng = np.random.RandomState(3)
    y = [np.array([v]) for v in 0.6 * rng.standard_normal(30)]
    fk = rssm.AuxiliaryPF(ssm=rssm.StochVol(mu=-1.0, rho=0.95, sigma=0.2), data=y)
    HipSMC = adapter.HipSMC()
    np.random.seed(5)
    want = particles.SMC(fk=fk, N=2000)
    want.run()
    got = {}
    for N in (1000, 2048, 3000):
        pf = HipSMC(fk=fk, N=N, seed=8)
        assert isinstance(pf, pa.SMC)
        got[N] = pf
        pf.run()
        assert abs(pf.logLt - want.logLt) < 0.5, (N, pf.logLt, want.logLt)
    assert got[1000]._fused and describe(got[1000]) == "k_filter_small"
    assert got[2048]._fused and describe(got[2048]) == "k_reduce2+k_ancestors2+k_propagate"
    assert got[3000]._fused and describe(got[3000]) == "k_reduce2+k_ancestors2+k_propagate"
    # ... the reference's AuxiliaryBootstrap (state_space_models.py:431-438) of a stock model: fused as well
    fkb = rssm.AuxiliaryBootstrap(ssm=rssm.StochVol(mu=-1.0, rho=0.95, sigma=0.2), data=y)
    np.random.seed(5)
    wantb = particles.SMC(fk=fkb, N=2000)
    wantb.run()
    for N, kern in ((1000, "k_filter_small"), (2048, "k_reduce2+k_ancestors2+k_propagate")):
        pf = HipSMC(fk=fkb, N=N, seed=8)
        assert isinstance(pf, pa.SMC) and pf._fused and pf.fk.isAPF and describe(pf) == kern
        pf.run()
        assert abs(pf.logLt - wantb.logLt) < 0.5, (N, pf.logLt, wantb.logLt)
    # ... and of the stock MVLinearGauss (kalman.py:348-361): the auxiliary weights in front of the flat step
    rk = ref["kalman"] if "kalman" in ref else __import__("particles.kalman", fromlist=["kalman"])
    model = rk.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=3)
    np.random.seed(2)
    x, ymv = model.simulate(12)
    fk = rssm.AuxiliaryPF(ssm=model, data=ymv)
    np.random.seed(6)
    want = particles.SMC(fk=fk, N=2000)
    want.run()
    pf = HipSMC(fk=fk, N=2000, seed=9)
    assert isinstance(pf, pa.SMC) and pf._fused and describe(pf).startswith("k_mv_aux+k_mv_aux_restate+")
    pf.run()
    assert abs(pf.logLt - want.logLt) < 0.5, (pf.logLt, want.logLt)


def test_deepcopy_clones_the_device_filter(ref):
    """copy.deepcopy of a device filter (what the reference's theta-level resampling does to every
    duplicated particle filter, smc_samplers.py:319-361): an independent filter in the same state --
    same particles, weights and evidence now, its own handle (destroying one leaves the other
    usable), its own noise from the next step on."""
    import copy
    import particles_amd as pa
    from particles_amd import kalman, state_space_models as ssm
    y = [np.array([0.3 * np.sin(t)]) for t in range(20)]
    for N in (300, 3000):                                   # one-launch filter / two-level step
        pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.LinearGauss(sigmaX=0.8, sigmaY=0.4, rho=0.9), data=y), N=N, seed=7)
        for _ in range(8):
            next(pf)
        cp = copy.deepcopy(pf)
        assert cp._f is not pf._f and cp._f.value != pf._f.value and cp.t == pf.t == 8
        assert np.array_equal(cp.X, pf.X) and np.array_equal(cp.wgts.lw, pf.wgts.lw) and cp.logLt == pf.logLt
        assert cp.summaries.ESSs == pf.summaries.ESSs and cp.summaries is not pf.summaries
        ref_run = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.LinearGauss(sigmaX=0.8, sigmaY=0.4, rho=0.9), data=y), N=N, seed=7)
        ref_run.run()
        for _ in range(12):
            next(pf)
            next(cp)
        assert pf.logLt == ref_run.logLt and np.array_equal(pf.X, ref_run.X)       # the source is undisturbed
        assert cp.logLt != pf.logLt and abs(cp.logLt - pf.logLt) < 3.0             # the copy drew its own noise
        del pf                                                                      # frees the source's filter only
        assert np.isfinite(cp.X).all() and len(cp.summaries.ESSs) == 20


def test_reference_smc2_runs_on_device_filters_under_install(ref):
    """The reference's own SMC2 (smc_samplers.py:1038-1167), unchanged, after adapter.install():
    alg_instance's particles.SMC(...) yields device filters, logG steps them one by one, the
    theta-level resampling deep-copies the duplicated ones (FancyList / all_distinct) and the PMCMC
    move builds fresh ones.  The run must complete, resample-move at least once, and its posterior
    and evidence must agree with the same algorithm on the reference's NumPy filters."""
    import particles_amd as pa
    from particles_amd import adapter
    particles, rk, dists = ref["particles"], ref["kalman"], ref["dists"]
    from particles import smc_samplers as ssp
    np.random.seed(4)
    model = rk.LinearGauss(sigmaX=1.0, sigmaY=0.4, rho=0.8)
    x, y = model.simulate(25)
    prior = dists.StructDist({"rho": dists.Uniform(a=0.3, b=0.99), "sigmaY": dists.Gamma(a=2.0, b=4.0)})

    def run(seed):
        np.random.seed(seed)
        # (an ADAPTIVE sequence of MCMC steps: not a form the device class implements, so the outer loop
        #  stays the reference's under install() -- only its inner filters change)
        fk = ssp.SMC2(ssm_cls=rk.LinearGauss, prior=prior, data=y, init_Nx=64, wastefree=False,
                      move=ssp.AdaptiveMCMCSequence(len_chain=3, adaptive=True))
        alg = particles.SMC(fk=fk, N=24, verbose=False)
        alg.run()
        return alg

    base = run(11)                                   # NumPy filters
    made = []
    real_init = pa.SMC.__init__

    def spy(self, *a, **kw):
        real_init(self, *a, **kw)
        made.append(self._fused)
    adapter.install()
    pa.SMC.__init__ = spy
    try:
        dev = run(11)
        outer_is_reference = not isinstance(dev, pa.SMC)
    finally:
        pa.SMC.__init__ = real_init
        adapter.uninstall()
    assert outer_is_reference                        # SMC2 itself is a Feynman-Kac object we do not reinterpret
    assert len(made) >= 24 and all(made)             # every inner filter is a fused device filter
    assert all(isinstance(pf, pa.SMC) for pf in dev.X.pfs)
    assert len({pf._f.value for pf in dev.X.pfs}) == len(dev.X.pfs) >= 24    # all distinct handles after the deep copies
    assert sum(dev.summaries.rs_flags) >= 1 and sum(base.summaries.rs_flags) >= 1
    assert np.isfinite(dev.logLt) and abs(dev.logLt - base.logLt) < 3.0, (dev.logLt, base.logLt)
    for k in ("rho", "sigmaY"):
        mb = np.average(base.X.theta[k], weights=base.W)
        md = np.average(dev.X.theta[k], weights=dev.W)
        assert abs(mb - md) < 0.25, (k, mb, md)


def test_reference_smc2_object_maps_onto_the_device_class(ref):
    """HipSMC(fk=<the reference's SMC2>, N=...) is the device class behind the outer SMC's attributes
    (X.theta, W, logLt, summaries, run / next), for the standard and for the waste-free (default) move;
    anything the device class does not implement exactly (an adaptive MCMC sequence, a user's model
    class) stays with the reference's loop."""
    from particles_amd import adapter, smc2
    particles, rk, dists, rssm = ref["particles"], ref["kalman"], ref["dists"], ref["ssm"]
    from particles import smc_samplers as ssp
    np.random.seed(4)
    x, y = rk.LinearGauss(sigmaX=1.0, sigmaY=0.4, rho=0.8).simulate(16)
    prior = dists.StructDist({"rho": dists.Uniform(a=0.3, b=0.99), "sigmaY": dists.Gamma(a=2.0, b=4.0)})
    HipSMC = adapter.HipSMC()
    fk = ssp.SMC2(ssm_cls=rk.LinearGauss, prior=prior, data=y, init_Nx=32, len_chain=3, wastefree=False)
    alg = HipSMC(fk=fk, N=16, seed=3)
    assert isinstance(alg, adapter.DeviceSMC2Run) and isinstance(alg._alg, smc2.SMC2)
    assert alg._alg.nmcmc == 2 and alg._alg.Nx == 32 and alg.N == 16
    alg.run()
    th = alg.X.theta
    assert th.dtype.names == ("rho", "sigmaY") and th.shape == (16,) and alg.t == 16
    assert np.all((th["rho"] > 0.3) & (th["rho"] < 0.99)) and np.all(th["sigmaY"] > 0)
    assert abs(alg.W.sum() - 1) < 1e-12 and np.isfinite(alg.logLt) and alg.cpu_time > 0
    s = alg.summaries
    assert len(s.ESSs) == len(s.logLts) == len(s.rs_flags) == 16 and s.logLts[-1] == alg.logLt
    assert not s.rs_flags[0] and all(0 < e <= 16 for e in s.ESSs)
    # one step at a time gives the same run (same seed)
    stp = HipSMC(fk=fk, N=16, seed=3)
    for _ in stp:
        pass
    assert stp.logLt == alg.logLt and np.array_equal(stp.X.theta, th) and stp.summaries.ESSs == s.ESSs
    # the reference's DEFAULT SMC2 -- the waste-free move (smc_samplers.py:669-684), len_chain states per chain
    wf = HipSMC(fk=ssp.SMC2(ssm_cls=rk.LinearGauss, prior=prior, data=y, init_Nx=32, len_chain=4), N=8, seed=5)
    assert isinstance(wf, adapter.DeviceSMC2Run) and wf._alg.wastefree and wf._alg.N == 32 and wf._alg.nmcmc == 3
    wf.run()
    assert wf.X.theta.shape == (32,) and abs(wf.W.sum() - 1) < 1e-12 and np.isfinite(wf.logLt) and wf.t == 16
    # not ours: an adaptive sequence of MCMC steps, a user's subclass of a stock model
    ad = HipSMC(fk=ssp.SMC2(ssm_cls=rk.LinearGauss, prior=prior, data=y, init_Nx=32, wastefree=False,
                            move=ssp.AdaptiveMCMCSequence(len_chain=3, adaptive=True)), N=16)
    assert isinstance(ad, particles.SMC) and not isinstance(ad, adapter.DeviceSMC2Run)

    class MyLG(rk.LinearGauss):
        pass
    assert adapter.adapt_smc2(ssp.SMC2(ssm_cls=MyLG, prior=prior, data=y, wastefree=False)) is None


def test_device_smc2_finalises_once_and_declines_what_it_cannot_batch(ref):
    """(a) The last evidence term is added exactly once: iterating to exhaustion and then calling run()
    (or run() twice) leaves logLt where one run() puts it -- the reference's run() on an exhausted SMC
    is a no-op.  (b) adapt_smc2 hands the reference's SMC2 to the device class only when the device class
    honours everything asked for: inner-filter options beyond resampling / ESSrmin (qmc, store_history ..)
    and model classes whose filters cannot be batched as islands with per-island parameters (the
    multivariate ones) stay with the reference's own loop; (c) the probe draws nothing from the caller's
    seeded numpy stream."""
    from particles_amd import adapter
    particles, rk, dists = ref["particles"], ref["kalman"], ref["dists"]
    from particles import smc_samplers as ssp
    np.random.seed(4)
    x, y = rk.LinearGauss(sigmaX=1.0, sigmaY=0.4, rho=0.8).simulate(10)
    prior = dists.StructDist({"rho": dists.Uniform(a=0.3, b=0.99), "sigmaY": dists.Gamma(a=2.0, b=4.0)})
    HipSMC = adapter.HipSMC()
    mk = lambda **kw: ssp.SMC2(ssm_cls=rk.LinearGauss, prior=prior, data=y, init_Nx=32, len_chain=3, wastefree=False, **kw)
    one = HipSMC(fk=mk(), N=16, seed=3)
    one.run()
    ll = one.logLt
    one.run()
    assert one.logLt == ll
    it = HipSMC(fk=mk(), N=16, seed=3)
    for _ in it:
        pass
    assert it.logLt == ll
    it.run()
    assert it.logLt == ll and it.summaries.logLts[-1] == ll
    # (b)
    assert adapter.adapt_smc2(mk(smc_options={"resampling": "stratified", "ESSrmin": 0.7})) is not None
    for opts in ({"qmc": True}, {"store_history": True}, {"collect": [particles.collectors.Moments()]}):
        fk = mk(smc_options=opts)
        assert adapter.adapt_smc2(fk) is None, opts
        alg = HipSMC(fk=fk, N=8)
        assert isinstance(alg, particles.SMC) and not isinstance(alg, adapter.DeviceSMC2Run)
    prior_mv = dists.StructDist({"alpha": dists.Uniform(a=0.1, b=0.6)})
    ymv = [np.zeros((1, 2)) for _ in range(4)]
    fk_mv = ssp.SMC2(ssm_cls=rk.MVLinearGauss_Guarniero_etal, prior=prior_mv, data=ymv, init_Nx=16, wastefree=False)
    assert adapter.adapt_smc2(fk_mv) is None
    # a model whose per-step term depends on a parameter the prior varies (Gordon_etal's d cos(e (t - 1)),
    # state_space_models.py:531-556, with `e` in the prior) cannot be batched as islands: the adapter says so on two
    # probe draws and the reference's loop runs it -- it used to pass a one-draw probe and then fail when the batch
    # was built
    rssm = ref["ssm"] if "ssm" in ref else __import__("particles.state_space_models", fromlist=["x"])
    yg = [np.array([v]) for v in np.random.RandomState(1).standard_normal(6)]
    prior_g = dists.StructDist({"e": dists.Uniform(a=1.0, b=1.4), "sigmaX": dists.Gamma(a=2.0, b=1.0)})
    fk_g = ssp.SMC2(ssm_cls=rssm.Gordon_etal, prior=prior_g, data=yg, init_Nx=16, len_chain=3, wastefree=False)
    assert adapter.adapt_smc2(fk_g) is None
    alg_g = HipSMC(fk=fk_g, N=6)
    assert isinstance(alg_g, particles.SMC) and not isinstance(alg_g, adapter.DeviceSMC2Run)
    prior_ok = dists.StructDist({"sigmaX": dists.Gamma(a=2.0, b=1.0)})              # (a parameter of the dynamics' scale only)
    assert adapter.adapt_smc2(ssp.SMC2(ssm_cls=rssm.Gordon_etal, prior=prior_ok, data=yg, init_Nx=16, len_chain=3,
                                       wastefree=False)) is not None
    # should the full batch be refused all the same, the run falls back to the reference's loop on an untouched stream
    import unittest.mock as um
    with um.patch.object(adapter, "DeviceSMC2Run", side_effect=ValueError("refused")):
        np.random.seed(5)
        fb = HipSMC(fk=mk(), N=8)
        assert isinstance(fb, particles.SMC)
        after = np.random.rand()
        np.random.seed(5)
        assert np.random.rand() == after
    # (c)
    np.random.seed(77)
    adapter.adapt_smc2(mk())
    after = np.random.rand()
    np.random.seed(77)
    assert np.random.rand() == after


def _pf_logLt(pf):
    return pf.logLt


def test_multismc_workers_return_pickled_device_filters(ref):
    """The REAL particles machinery (utils.multiplexer, loky worker processes) with HipSMC in the workers
    (adapter.multiSMC): nprocs = 2 worker processes build device filters, run them, and hand the finished SMC objects
    back PICKLED (SMC.__getstate__: smc_filter_save_state) -- core.py:415-428, utils.py:178-186.  What arrives is a live
    device filter again: its particles, weights and summaries are those of the run, equal to the same seeded runs done
    in this process."""
    import pickle
    import particles_amd as pa
    from particles_amd import adapter
    rk, rssm = ref["kalman"], ref["ssm"]
    np.random.seed(5)
    model = rk.LinearGauss(sigmaX=1.0, sigmaY=0.4, rho=0.9)
    x, y = model.simulate(12)
    fk = rssm.Bootstrap(ssm=model, data=y)
    np.random.seed(77)
    par = adapter.multiSMC(fk=fk, N=3000, nruns=4, nprocs=2)
    assert [r["run"] for r in par] == [0, 1, 2, 3]
    for a in par:
        pa_ = a["output"]
        assert isinstance(pa_, pa.SMC) and pa_._fused and pa_.t == 12
        pb = pa.SMC(fk=adapter.adapt(fk), N=3000, seed=pa_.seed)             # the same Philox key, run here
        pb.run()
        assert pa_.logLt == pb.logLt and np.array_equal(np.array(pa_.X), np.array(pb.X))
        assert np.array_equal(np.array(pa_.wgts.W), np.array(pb.wgts.W))
        assert list(pa_.summaries.logLts) == list(pb.summaries.logLts)
    assert len({r["output"].logLt for r in par}) == 4                    # distinct seeds, distinct runs
    # out_func runs in the worker: only its value travels
    lls = adapter.multiSMC(fk=fk, N=3000, nruns=4, nprocs=2, out_func=_pf_logLt)
    assert all(isinstance(r["output"], float) for r in lls) and len({r["output"] for r in lls}) == 4
    # and a pickled filter is a checkpoint: it resumes where it stood
    q = pa.SMC(fk=adapter.adapt(fk), N=3000, seed=9)
    for _ in range(5):
        next(q)
    z = pickle.loads(pickle.dumps(q))
    q.run(); z.run()
    assert q.logLt == z.logLt and np.array_equal(np.array(q.X), np.array(z.X))
