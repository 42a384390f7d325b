"""Counterpart of ``particles.state_space_models`` for the hot path:
``StateSpaceModel`` (:172-296), ``Bootstrap`` (:299-349), ``GuidedPF``
(:352-398) and ``StochVol`` (:446-499).

A model is still declared the reference's way -- ``PX0`` / ``PX`` / ``PY`` (and
``proposal0`` / ``proposal``) returning ``ProbDist`` objects of
``particles_amd.distributions`` -- so those methods work on arrays as in the
reference.  In addition the model classes of the closed family the fused HIP
step loop knows (``kalman.LinearGauss``, ``StochVol``, ``kalman.MVLinearGauss``)
export ``_device_params()``; ``Bootstrap`` / ``GuidedPF`` pass it on to
``particles_amd.SMC`` which then runs the whole time loop on the device.
"""
import numpy as np

from . import _lib
from . import distributions as dists
from .core import FeynmanKac


class StateSpaceModel:
    """Base class for state-space models (state_space_models.py:172-296)."""

    def __init__(self, **kwargs):
        if hasattr(self, "default_params"):
            self.__dict__.update(self.default_params)
        self.__dict__.update(kwargs)

    def _error_msg(self, method):
        return "method " + method + " not implemented in class%s" % self.__class__.__name__

    def PX0(self):
        "Law of X_0 at time 0"
        raise NotImplementedError(self._error_msg("PX0"))

    def PX(self, t, xp):
        "Law of X_t at time t, given X_{t-1} = xp"
        raise NotImplementedError(self._error_msg("PX"))

    def PY(self, t, xp, x):
        """Conditional distribution of Y_t, given the states."""
        raise NotImplementedError(self._error_msg("PY"))

    def proposal0(self, data):
        raise NotImplementedError(self._error_msg("proposal0"))

    def proposal(self, t, xp, data):
        raise NotImplementedError(self._error_msg("proposal"))

    def simulate_given_x(self, x):
        lag_x = [None] + x[:-1]
        return [self.PY(t, xp, x).rvs(size=1) for t, (xp, x) in enumerate(zip(lag_x, x))]

    def simulate(self, T):
        """Simulate state and observation processes (state_space_models.py:278-296):
        lists x, y of length T (draws come from the device Philox stream)."""
        x = []
        for t in range(T):
            law_x = self.PX0() if t == 0 else self.PX(t, x[-1])
            x.append(law_x.rvs(size=1))
        y = self.simulate_given_x(x)
        return x, y

    def _device_params(self, fk_kind):
        """(model kind, dx, dy, params (16,) or None, matrices) for the fused
        step loop, or None when the model is outside the closed family."""
        return None


class Bootstrap(FeynmanKac):
    """Bootstrap Feynman-Kac formalism of a state-space model
    (state_space_models.py:299-349)."""

    _fk_kind = _lib.FK_BOOTSTRAP

    def __init__(self, ssm=None, data=None):
        self.ssm = ssm
        self.data = data
        self.du = self.ssm.PX0().dim

    @property
    def T(self):
        return 0 if self.data is None else len(self.data)

    def M0(self, N):
        return self.ssm.PX0().rvs(size=N)

    def M(self, t, xp):
        return self.ssm.PX(t, xp).rvs(size=xp.shape[0])

    def logG(self, t, xp, x):
        return self.ssm.PY(t, xp, x).logpdf(self.data[t])

    def Gamma0(self, u):
        return self.ssm.PX0().ppf(u)                       # state_space_models.py:335-336

    def Gamma(self, t, xp, u):
        return self.ssm.PX(t, xp).ppf(u)                   # :338-340

    def logpt(self, t, xp, x):
        """PDF of X_t|X_{t-1}=xp"""
        return self.ssm.PX(t, xp).logpdf(x)

    def _device_model(self):
        """Parameters of the fused device loop, or None.  The fused kernels implement the STOCK
        model: it is chosen only when neither this Feynman-Kac class nor the state-space model
        overrides one of the methods the kernels stand for -- a user subclass that redefines
        ``PY`` (or ``logG``, ``time_to_resample``, ...) must run its own code, through the
        template-method path (the reference's normal way to customise a model)."""
        base = GuidedPF if isinstance(self, GuidedPF) else Bootstrap
        cls = type(self)
        for name in ("M0", "M", "logG", "time_to_resample", "done"):
            if getattr(cls, name) is not getattr(base, name):
                return None
        scls = type(self.ssm)
        owner = next((c for c in scls.__mro__ if "_device_params" in vars(c)), None)
        if owner is None:
            return None
        for name in ("PX0", "PX", "PY", "proposal0", "proposal"):
            if getattr(scls, name, None) is not getattr(owner, name, None):
                return None
        return self.ssm._device_params(self._fk_kind)


class GuidedPF(Bootstrap):
    """Guided filter for a state-space model with ``proposal0`` / ``proposal``
    (state_space_models.py:352-398)."""

    _fk_kind = _lib.FK_GUIDED

    def M0(self, N):
        return self.ssm.proposal0(self.data).rvs(size=N)

    def M(self, t, xp):
        return self.ssm.proposal(t, xp, self.data).rvs(size=xp.shape[0])

    def Gamma0(self, u):
        return self.ssm.proposal0(self.data).ppf(u)        # :394-395

    def Gamma(self, t, xp, u):
        return self.ssm.proposal(t, xp, self.data).ppf(u)  # :397-398

    def logG(self, t, xp, x):
        if t == 0:
            return (self.ssm.PX0().logpdf(x)
                    + self.ssm.PY(0, xp, x).logpdf(self.data[0])
                    - self.ssm.proposal0(self.data).logpdf(x))
        return (self.ssm.PX(t, xp).logpdf(x)
                + self.ssm.PY(t, xp, x).logpdf(self.data[t])
                - self.ssm.proposal(t, xp, self.data).logpdf(x))


class APFMixin:
    def logeta(self, t, x):
        return self.ssm.logeta(t, x, self.data)


class AuxiliaryPF(GuidedPF, APFMixin):
    """Auxiliary particle filter (state_space_models.py:406-428); ``ssm`` must implement
    ``proposal0``, ``proposal`` and ``logeta``.  For the stock ``StochVol`` and N <= 1024 the
    whole filter -- auxiliary weights included -- runs fused on the device (one launch for the
    T-loop, ``k_filter_small``); otherwise the template-method step with device operators (the
    auxiliary weights are one more ``Weights.add`` / weighted log-mean-exp)."""

    _fk_kind = _lib.FK_APF

    def _device_model(self):
        base = GuidedPF._device_model(self)
        if base is None or not base.get("apf") or type(self).logeta is not APFMixin.logeta:
            return None
        owner = next((c for c in type(self.ssm).__mro__ if "_device_params" in vars(c)), None)
        if getattr(type(self.ssm), "logeta", None) is not getattr(owner, "logeta", None):
            return None
        return base


class AuxiliaryBootstrap(Bootstrap, APFMixin):
    """APF whose proposal is the transition kernel (state_space_models.py:431-438)."""

    def _device_model(self):
        return None


class StochVol(StateSpaceModel):
    r"""Univariate stochastic volatility model (state_space_models.py:446-473).

    X_0 ~ N(mu, sigma^2/(1-rho^2)); X_t = mu + rho (X_{t-1}-mu) + sigma U_t;
    Y_t | X_t ~ N(0, exp(X_t)).
    """
    default_params = {"mu": -1.02, "rho": 0.9702, "sigma": 0.178}

    def sig0(self):
        return self.sigma / np.sqrt(1.0 - self.rho ** 2)

    def PX0(self):
        return dists.Normal(loc=self.mu, scale=self.sig0())

    def EXt(self, xp):
        return (1.0 - self.rho) * self.mu + self.rho * xp

    def PX(self, t, xp):
        return dists.Normal(loc=self.EXt(xp), scale=self.sigma)

    def PY(self, t, xp, x):
        return dists.Normal(loc=0.0, scale=np.exp(0.5 * x))

    # Pitt & Shephard's proposal and auxiliary function (state_space_models.py:475-498)
    def _xhat(self, xst, sig, yt):
        return xst + 0.5 * sig ** 2 * (yt ** 2 * np.exp(-xst) - 1.0)

    def proposal0(self, data):
        return dists.Normal(loc=self._xhat(0.0, self.sig0(), data[0]), scale=self.sig0())

    def proposal(self, t, xp, data):
        return dists.Normal(loc=self._xhat(self.EXt(xp), self.sigma, data[t]), scale=self.sigma)

    def logeta(self, t, x, data):
        xst = self.EXt(x)
        xstmmu = xst - self.mu
        xhat = self._xhat(xst, self.sigma, data[t + 1])
        xhatmmu = xhat - self.mu
        return 0.5 / self.sigma ** 2 * (xhatmmu ** 2 - xstmmu ** 2) - 0.5 * data[
            t + 1] ** 2 * np.exp(-xst) * (1.0 + xstmmu)

    def _device_params(self, fk_kind):
        p = np.zeros(_lib.PARAM_STRIDE)
        p[:5] = [self.mu, self.rho, self.sigma, self.sig0(), (1.0 - self.rho) * self.mu]
        if fk_kind != _lib.FK_BOOTSTRAP:          # Pitt & Shephard's proposal / logeta (:475-498)
            p[5:10] = [np.log(self.sigma), np.log(self.sig0()), 0.5 * self.sigma ** 2,
                       0.5 * self.sig0() ** 2, 0.5 / self.sigma ** 2]
        return dict(kind=_lib.MODEL_STOCHVOL, dx=1, dy=1, params=p, apf=True)


class StochVolLeverage(StochVol):
    r"""Stochastic volatility with leverage (state_space_models.py:501-541): the innovations
    of X_t and Y_t have correlation phi, i.e. Y_t | X_{t-1:t} ~ N(s phi z, s^2 (1 - phi^2)),
    s = exp(x_t / 2), z = [x_t - mu - rho (x_{t-1} - mu)] / sigma."""
    default_params = {"mu": -1.02, "rho": 0.9702, "sigma": 0.178, "phi": 0.0}

    def PY(self, t, xp, x):
        if t == 0:
            u = (x - self.mu) / self.sig0()
        else:
            u = (x - self.EXt(xp)) / self.sigma
        std_x = np.exp(0.5 * x)
        return dists.Normal(loc=std_x * self.phi * u, scale=std_x * np.sqrt(1.0 - self.phi ** 2))

    def _device_params(self, fk_kind):
        if fk_kind != _lib.FK_BOOTSTRAP:
            return None
        p = np.zeros(_lib.PARAM_STRIDE)
        p[:7] = [self.mu, self.rho, self.sigma, self.sig0(), (1.0 - self.rho) * self.mu, self.phi,
                 np.sqrt(1.0 - self.phi ** 2)]
        return dict(kind=_lib.MODEL_SVLEVERAGE, dx=1, dy=1, params=p)


class Gordon_etal(StateSpaceModel):
    r"""Toy example of Gordon et al (1993) (state_space_models.py:546-577).

    X_0 ~ N(0, 2^2); X_t = b X_{t-1} + c X_{t-1}/(1+X_{t-1}^2) + d cos(e (t-1)) + sigmaX V_t;
    Y_t | X_t ~ N(a X_t^2, 1).
    """
    default_params = {"a": 0.05, "b": 0.5, "c": 25.0, "d": 8.0, "e": 1.2, "sigmaX": 3.162278}

    def PX0(self):
        return dists.Normal(scale=2.0)

    def PX(self, t, xp):
        return dists.Normal(loc=self.b * xp + self.c * xp / (1.0 + xp ** 2)
                            + self.d * np.cos(self.e * (t - 1)), scale=self.sigmaX)

    def PY(self, t, xp, x):
        return dists.Normal(loc=self.a * x ** 2)

    def _device_params(self, fk_kind):
        if fk_kind != _lib.FK_BOOTSTRAP:
            return None
        p = np.zeros(_lib.PARAM_STRIDE)
        p[[0, 1, 2, 3, 5]] = [self.b, self.sigmaX, self.c, 2.0, self.a]
        # the time-dependent term, with the host's cos (as PX evaluates it)
        aux = lambda T: self.d * np.cos(self.e * (np.arange(T) - 1))
        return dict(kind=_lib.MODEL_GORDON, dx=1, dy=1, params=p, aux=aux)


class DiscreteCox(StateSpaceModel):
    r"""A discrete Cox model (state_space_models.py:611-630).

    Y_t | X_t = x ~ Poisson(e^x);  X_t = mu + phi (X_{t-1} - mu) + U_t, U_t ~ N(0, sigma^2);
    X_0 ~ N(mu, sigma^2 / (1 - phi^2)).
    """
    default_params = {"mu": 0.0, "sigma": 1.0, "phi": 0.95}

    def PX0(self):
        return dists.Normal(loc=self.mu, scale=self.sigma / np.sqrt(1.0 - self.phi ** 2))

    def PX(self, t, xp):
        return dists.Normal(loc=self.mu + self.phi * (xp - self.mu), scale=self.sigma)

    def PY(self, t, xp, x):
        return dists.Poisson(rate=np.exp(x))

    def _device_params(self, fk_kind):
        if fk_kind != _lib.FK_BOOTSTRAP:
            return None
        p = np.zeros(_lib.PARAM_STRIDE)
        p[:4] = [self.mu, self.phi, self.sigma, self.sigma / np.sqrt(1.0 - self.phi ** 2)]

        def aux(y):
            # the data-only term of scipy's poisson._logpmf; counts outside the support get
            # +inf so that every particle's increment is -inf, as rv_discrete.logpmf returns
            from scipy.special import gammaln
            y = np.asarray(y, dtype=np.float64).reshape(-1)
            ok = (y >= 0) & (np.floor(y) == y)
            return np.where(ok, gammaln(np.where(ok, y, 0.0) + 1.0), np.inf)
        return dict(kind=_lib.MODEL_DISCRETECOX, dx=1, dy=1, params=p, aux_from_data=aux)


class ThetaLogistic(StateSpaceModel):
    r"""Theta-logistic model (state_space_models.py:657-683).

    X_0 ~ N(0,1); X_t = X_{t-1} + tau0 - tau1 exp(tau2 X_{t-1}) + U_t, U_t ~ N(0, sigmaX^2);
    Y_t = X_t + V_t, V_t ~ N(0, sigmaY^2).
    """
    default_params = {"tau0": 0.15, "tau1": 0.12, "tau2": 0.1, "sigmaX": 0.47, "sigmaY": 0.39}

    def PX0(self):
        return dists.Normal(loc=0.0, scale=1.0)

    def PX(self, t, xp):
        return dists.Normal(loc=xp + self.tau0 - self.tau1 * np.exp(self.tau2 * xp),
                            scale=self.sigmaX)

    def PY(self, t, xp, x):
        return dists.Normal(loc=x, scale=self.sigmaY)

    def _device_params(self, fk_kind):
        if fk_kind != _lib.FK_BOOTSTRAP:
            return None
        p = np.zeros(_lib.PARAM_STRIDE)
        p[:7] = [self.tau0, self.sigmaX, self.sigmaY, 1.0, np.log(self.sigmaY), self.tau1,
                 self.tau2]
        return dict(kind=_lib.MODEL_THETALOGISTIC, dx=1, dy=1, params=p)
