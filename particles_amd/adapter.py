"""Reference-side adapter: run objects built with the REAL ``particles`` package on the device.

The reference's injection seams take classes and registries, not this package's
mirrors: ``PMMH(smc_cls=...)`` (mcmc.py:372,419,439), ``particles.SMC`` looked up at call time by
``SMC2.alg_instance`` (smc_samplers.py:1122-1127), ``rs.rs_funcs`` (resampling.py:445-481).  This
module plugs into exactly those:

    import particles, particles_amd.adapter as hip
    pmmh = particles.mcmc.PMMH(..., smc_cls=hip.HipSMC())          # fused device filter per theta
    alg = hip.HipSMC()(fk=particles.smc_samplers.SMC2(..., wastefree=False), N=...)   # SMC^2 on the device class
    hip.register_into(particles.resampling)                         # 'systematic_hip', ...
    hip.install()      # particles.SMC -> HipSMC, for callers that name particles.SMC themselves

``adapt(fk)`` maps a reference Feynman-Kac object -- ``Bootstrap`` / ``GuidedPF`` (/ APF) of a STOCK
state-space model, recognised by exact class, parameters read from its ``__dict__`` -- onto the
same-named classes here; anything else (user subclasses, other models) is left alone and
``HipSMC`` then IS the reference's ``particles.SMC`` (it subclasses it), i.e. the NumPy path.
``particles`` is imported lazily: nothing here is needed, or loaded, when it is absent.
"""
import numpy as np

from . import _lib
from . import kalman
from . import resampling as _rs
from . import state_space_models as ssm
from .core import SMC as _DeviceSMC

# reference class name -> (our class, constructor arguments read from the instance)
_SSM_TABLE = {
    "LinearGauss": (kalman.LinearGauss, ("sigmaX", "sigmaY", "rho", "sigma0")),
    "MVLinearGauss": (kalman.MVLinearGauss, ("F", "G", "covX", "covY", "mu0", "cov0")),
    "MVLinearGauss_Guarniero_etal": (kalman.MVLinearGauss, ("F", "G", "covX", "covY", "mu0", "cov0")),
    "StochVol": (ssm.StochVol, ("mu", "rho", "sigma")),
    "StochVolLeverage": (ssm.StochVolLeverage, ("mu", "rho", "sigma", "phi")),
    "Gordon_etal": (ssm.Gordon_etal, ("a", "b", "c", "d", "e", "sigmaX")),
    "ThetaLogistic": (ssm.ThetaLogistic, ("tau0", "tau1", "tau2", "sigmaX", "sigmaY")),
    "DiscreteCox": (ssm.DiscreteCox, ("mu", "sigma", "phi")),
    # (no fused descriptor: the template-method step on device operators)
    "BearingsOnly": (ssm.BearingsOnly, ("sigmaX", "sigmaY", "x0")),
    "MVStochVol": (ssm.MVStochVol, ("mu", "covX", "corY", "F")),
}
_FK_TABLE = {"Bootstrap": ssm.Bootstrap, "GuidedPF": ssm.GuidedPF,
             "AuxiliaryPF": ssm.AuxiliaryPF, "AuxiliaryBootstrap": ssm.AuxiliaryBootstrap}


def _stock(obj, modules):
    """The reference's own class of that name, if `obj` is EXACTLY an instance of it."""
    import importlib
    name = type(obj).__name__
    for m in modules:
        try:
            cls = getattr(importlib.import_module(m), name, None)
        except ImportError:
            cls = None
        if cls is not None and type(obj) is cls:
            return name
    return None


def adapt_ssm(model):
    """particles.*.<StockModel> instance -> the same-named model of this package, or None."""
    name = _stock(model, ("particles.kalman", "particles.state_space_models"))
    if name is None or name not in _SSM_TABLE:
        return None
    cls, keys = _SSM_TABLE[name]
    kw = {k: getattr(model, k) for k in keys if hasattr(model, k)}
    return cls(**kw)


def adapt(fk):
    """particles.state_space_models.{Bootstrap, GuidedPF, AuxiliaryPF, AuxiliaryBootstrap} of a
    stock model -> the same Feynman-Kac object of this package (runs fused where the family
    allows, on device operators otherwise); None if it is anything else."""
    name = _stock(fk, ("particles.state_space_models",))
    if name is None or name not in _FK_TABLE:
        return None
    model = adapt_ssm(fk.ssm)
    if model is None:
        return None
    return _FK_TABLE[name](ssm=model, data=fk.data)


class _Theta:
    """What ``alg.X`` of an outer SMC over SMC2 exposes (smc_samplers.py ThetaParticles): ``theta`` as a
    structured array, ``lpost``, ``N``, the inner filters' size."""

    def __init__(self, run):
        self._run = run

    @property
    def theta(self):
        d = self._run._alg.theta
        out = np.empty(self._run._alg.N, dtype=[(k, float) for k in d])     # (waste-free: N x len_chain of them)
        for k, v in d.items():
            out[k] = v
        return out

    @property
    def N(self):
        return self._run._alg.N

    @property
    def Nx(self):
        return self._run._alg.Nx

    @property
    def lpost(self):
        a = self._run._alg
        with np.errstate(all="ignore"):
            return np.asarray(a.prior.logpdf(a.theta), dtype=float) + a._evidences(a.pf)


class _ThetaWeights:
    def __init__(self, run):
        self._run = run

    @property
    def lw(self):
        return self._run._alg.lw

    @property
    def W(self):
        return self._run._alg.W

    @property
    def ESS(self):
        W = self.W
        return 1.0 / np.sum(W * W)


class _Summ:
    def __init__(self, alg):
        self.ESSs, self.logLts, self.rs_flags = list(alg.ESSs), list(alg.logLts), list(alg.rs_flags)


class DeviceSMC2Run:
    """``particles.SMC(fk=SMC2(ssm_cls, prior, data, init_Nx, ar_to_increase_Nx, len_chain[, wastefree=False]), N=...)``
    (smc_samplers.py:1038-1167 under core.py:200-409) on the device class ``particles_amd.smc2.SMC2``:
    all N filters are islands of one device filter, the theta level lives on the device, a
    resample-move re-runs candidate batches.  Same algorithm as the reference's standard (not
    waste-free) resample-move: ``len_chain - 1`` random-walk Metropolis steps calibrated on the weighted
    theta-particles (smc_samplers.py:617-632), systematic theta-resampling at ESS < ESSrmin N, exchange
    step when the acceptance rate falls below ``ar_to_increase_Nx``.  Attributes of the outer SMC that
    callers read: ``X.theta``, ``W``, ``wgts``, ``logLt``, ``t``, ``N``, ``fk``, ``summaries`` (ESSs,
    logLts, rs_flags), ``cpu_time``; ``run()`` / ``next()`` / iteration."""

    def __init__(self, fk, ssm_cls, fk_cls, N=100, ESSrmin=0.5, resampling="systematic", seed=None, **kw):
        from . import smc2
        if resampling != "systematic":
            raise ValueError("SMC^2 on the device resamples the theta level with the systematic scheme")
        self.fk, self.N, self.ESSrmin = fk, N, ESSrmin
        opts = {k: v for k, v in (fk.smc_options or {}).items() if k in ("resampling", "ESSrmin")}
        self._alg = smc2.SMC2(ssm_cls=ssm_cls, prior=fk.prior, data=fk.data, init_Nx=fk.init_Nx, N=N, fk_cls=fk_cls,
                              ESSrmin=ESSrmin, nmcmc=fk.move.nsteps, ar_to_increase_Nx=fk.ar_to_increase_Nx,
                              smc_options=opts, seed=seed, sync_every=kw.pop("sync_every", 16),
                              wastefree=bool(fk.wastefree), len_chain=fk.move.nsteps + 1)
        self.X, self.wgts = _Theta(self), _ThetaWeights(self)
        self.cpu_time = 0.0
        self.rs_flag = False

    t = property(lambda self: self._alg.t)
    logLt = property(lambda self: self._alg.logLts[-1] if self._alg.logLts and self._alg.t < self._alg.T
                     else self._alg.logLt)
    W = property(lambda self: self._alg.W)
    summaries = property(lambda self: _Summ(self._alg))

    def run(self):
        import time
        t0 = time.perf_counter()
        self._alg.run()
        self.cpu_time = time.perf_counter() - t0

    def __next__(self):
        a = self._alg
        if a.t >= a.T:
            raise StopIteration
        keep, a.sync_every = a.sync_every, 1
        T_keep, t0 = a.T, a.t
        try:                                   # one time step: the loop of run() limited to t0 + 1
            a.pf.step_async(1)
            lw, stop, done, ess = a._theta_state(a.pf)
            t_new = stop if stop else done
            a.ESSs.extend(ess[a.t:t_new].tolist())
            lm = a._theta_logmeans(a.pf)
            a.logLts.extend((a.logLt + lm[a.t:t_new] - a._log_mean(a._lw_at_reset)).tolist())
            a.lw, a.t = lw, t_new
            a.pf.t = a.pf._n = t_new
            a.pf._invalidate()
            self.rs_flag = bool(stop)
            if stop:
                a._resample_move()
            a._finalise()                      # (the last evidence term: once, whoever reaches t = T first)
        finally:
            a.sync_every = keep

    next = __next__

    def __iter__(self):
        return self


def adapt_smc2(fk):
    """(our model class, our Feynman-Kac class) if `fk` is the reference's SMC2 in a form the device
    class implements exactly -- stock model class, Bootstrap / GuidedPF inner filters, random-walk
    Metropolis steps in the waste-free sequence (the reference's default) or the standard one -- else
    None (the reference's outer loop then runs it; with ``install()`` its inner filters are still device
    filters)."""
    try:
        from particles import smc_samplers as ssp
        from particles import state_space_models as rssm
    except ImportError:
        return None
    if type(fk) is not ssp.SMC2:
        return None
    if fk.wastefree:        # the reference's default: MCMCSequenceWF (smc_samplers.py:669-684) of random-walk steps
        if type(fk.move) is not ssp.MCMCSequenceWF or type(fk.move.mcmc) is not ssp.ArrayRandomWalk:
            return None
    elif type(fk.move) is not ssp.AdaptiveMCMCSequence or fk.move.adaptive or type(fk.move.mcmc) is not ssp.ArrayRandomWalk:
        return None
    name = fk.ssm_cls.__name__
    import importlib
    stock = None
    for m in ("particles.kalman", "particles.state_space_models"):
        if getattr(importlib.import_module(m), name, None) is fk.ssm_cls:
            stock = name
    if stock is None or stock not in _SSM_TABLE:
        return None
    ours, keys = _SSM_TABLE[stock]
    if fk.fk_cls is rssm.Bootstrap:
        fkc = ssm.Bootstrap
    elif fk.fk_cls is rssm.GuidedPF:
        fkc = ssm.GuidedPF
    else:
        return None
    # the inner filters' options: the device class forwards `resampling` and `ESSrmin` and nothing else --
    # anything more (qmc, store_history, collectors ...) belongs to the reference's own loop
    if set(fk.smc_options or {}) - {"collect", "resampling", "ESSrmin"}:
        return None
    if (fk.smc_options or {}).get("collect") not in (None, "off"):
        return None
    # ... and the inner filters must be batchable as islands with per-island parameters: a univariate model of
    # the fused family (a probe instance at the prior's first draw says which)
    # (TWO probes at different prior draws, as islands of one throw-away filter: a per-step term that depends on a
    #  parameter the prior varies -- Gordon_etal with `d` or `e` in the prior -- and an inner resampling scheme the
    #  fused step does not run are refused HERE, by the very checks the batch will meet, core.py _create_filter)
    state = np.random.get_state()                  # (the probes must not cost the caller's seeded stream a draw)
    try:
        probe = fk.prior.rvs(size=2)
        fks = [fkc(ssm=ours(**{k: float(np.ravel(probe[k])[i]) for k in probe.dtype.names}), data=fk.data)
               for i in range(2)]
        dm = fks[0]._device_model()
        if dm is None or dm.get("kind") == _lib.MODEL_MVLINGAUSS:
            return None
        opts = {k: v for k, v in (fk.smc_options or {}).items() if k in ("resampling", "ESSrmin")}
        from .core import SMC as _DevSMC
        trial = _DevSMC(fk=fks, N=max(2, min(int(fk.init_Nx), 64)), collect="off", seed=0, **opts)
        if not getattr(trial, "_fused", False):
            return None
        del trial
    except Exception:
        return None
    finally:
        np.random.set_state(state)
    return ours, fkc


_HIP_SMC = None


def HipSMC():
    """The class to hand to the reference: ``class HipSMC(particles.SMC)`` whose constructor returns
    a device-resident ``particles_amd.SMC`` when ``adapt(fk)`` succeeds (same attributes and
    iterator protocol: run(), next(), logLt, loglt, X, A, wgts, summaries, cpu_time) and an
    ordinary ``particles.SMC`` otherwise."""
    global _HIP_SMC
    if _HIP_SMC is None:
        import particles

        base = getattr(particles, "_smc_before_hip", particles.SMC)

        class HipSMC(base):
            def __new__(cls, fk=None, **kw):
                two = adapt_smc2(fk) if fk is not None else None
                if two is not None and not kw.get("qmc") and not kw.get("store_history") \
                        and kw.get("collect") in (None, "off") and not kw.get("verbose") \
                        and kw.get("resampling", "systematic") == "systematic":     # (the device theta level's scheme)
                    kw2 = {k: v for k, v in kw.items() if k in ("N", "ESSrmin", "resampling", "seed")}
                    # adapt_smc2 ran the batchability checks on two probe islands; should the full batch still be
                    # refused (a ValueError while the device run is being BUILT: nothing has been stepped), the
                    # reference's own loop runs it, on the random stream as the caller seeded it
                    state = np.random.get_state()
                    try:
                        return DeviceSMC2Run(fk, two[0], two[1], **kw2)
                    except ValueError:
                        np.random.set_state(state)
                        return super().__new__(cls)
                mine = adapt(fk) if fk is not None else None
                if mine is None or kw.get("qmc"):
                    return super().__new__(cls)            # the reference's own path
                kw.pop("qmc", None)
                return _DeviceSMC(fk=mine, **kw)           # not an instance of cls: __init__ is skipped

        _HIP_SMC = HipSMC
    return _HIP_SMC


class _HipRun:
    """The worker functor of the reference's multiSMC (core.py:415-423 ``_picklable_f``) with HipSMC where it names
    ``SMC``: picklable, so the reference's loky workers (utils.py:178-186) can receive it -- each worker process then
    builds device filters of its own and sends the finished ``SMC`` objects back PICKLED (``SMC.__getstate__``: the
    filter's device state as one host buffer), exactly as the reference returns its NumPy-backed ones."""

    def __init__(self, fun=None):
        self.fun = fun

    def __call__(self, **kwargs):
        pf = HipSMC()(**kwargs)
        pf.run()
        return pf if self.fun is None else self.fun(pf)


def multiSMC(nruns=10, nprocs=0, out_func=None, collect=None, **args):
    """``particles.multiSMC`` (core.py:431-518) with every run a HipSMC: the reference's own ``utils.multiplexer`` --
    cartesian products of list / dict arguments, one seed per run, ``nprocs`` loky worker processes -- drives it, the
    result is the reference's list of dicts (``'output'``: the finished filter, or ``out_func`` of it)."""
    from particles import utils
    return utils.multiplexer(f=_HipRun(out_func), nruns=nruns, nprocs=nprocs, seeding=True,
                             protected_args={"collect": collect}, **args)


def install():
    """particles.SMC := HipSMC (callers that look ``particles.SMC`` up at call time, e.g.
    SMC2.alg_instance, smc_samplers.py:1122-1127).  Undo with ``uninstall()``."""
    import particles
    if not hasattr(particles, "_smc_before_hip"):
        particles._smc_before_hip = particles.SMC
    particles.SMC = HipSMC()


def uninstall():
    import particles
    if hasattr(particles, "_smc_before_hip"):
        particles.SMC = particles._smc_before_hip
        del particles._smc_before_hip


def register_into(rs_module, suffix="_hip", override=False):
    """Add the device resampling schemes to the reference's registry (resampling.py:445-481):
    ``rs_funcs['systematic_hip']`` etc. -- or, with override=True, under the reference's own names.
    The uniforms are drawn from numpy's global generator in the reference's order, so a seeded
    run consumes the same stream as with the reference's functions."""
    names = ("systematic", "stratified", "multinomial", "residual", "killing", "ssp")
    added = []
    for n in names:
        f = _rs.rs_funcs[n]

        def wrapped(W, M=None, _f=f):
            return _f(np.asarray(W), M=M)

        wrapped.__name__ = n + ("" if override else suffix)
        wrapped.__doc__ = "device implementation of resampling.%s (particles_amd)" % n
        rs_module.rs_funcs[wrapped.__name__] = wrapped
        added.append(wrapped.__name__)
    return added
