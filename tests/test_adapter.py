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
