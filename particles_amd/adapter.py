"""Reference-side adapter: run objects built with the REAL ``particles`` package on the device.

The reference's injection seams take classes and registries, not this package's
mirrors: ``PMMH(smc_cls=...)`` (mcmc.py:372,419,439), ``particles.SMC`` looked up at call time by
``SMC2.alg_instance`` (smc_samplers.py:1122-1127), ``rs.rs_funcs`` (resampling.py:445-481).  This
module plugs into exactly those:

    import particles, particles_amd.adapter as hip
    pmmh = particles.mcmc.PMMH(..., smc_cls=hip.HipSMC())          # fused device filter per theta
    hip.register_into(particles.resampling)                         # 'systematic_hip', ...
    hip.install()      # particles.SMC -> HipSMC, for callers that name particles.SMC themselves

``adapt(fk)`` maps a reference Feynman-Kac object -- ``Bootstrap`` / ``GuidedPF`` (/ APF) of a STOCK
state-space model, recognised by exact class, parameters read from its ``__dict__`` -- onto the
same-named classes here; anything else (user subclasses, other models) is left alone and
``HipSMC`` then IS the reference's ``particles.SMC`` (it subclasses it), i.e. the NumPy path.
``particles`` is imported lazily: nothing here is needed, or loaded, when it is absent.
"""
import numpy as np

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
                mine = adapt(fk) if fk is not None else None
                if mine is None or kw.get("qmc"):
                    return super().__new__(cls)            # the reference's own path
                kw.pop("qmc", None)
                return _DeviceSMC(fk=mine, **kw)           # not an instance of cls: __init__ is skipped

        _HIP_SMC = HipSMC
    return _HIP_SMC


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
