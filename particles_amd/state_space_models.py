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


def _gauss(loc=0.0, scale=1.0):
    return dists.Normal(loc=loc, scale=scale)


class StateSpaceModel:
    """Base class for state-space models (state_space_models.py:172-296): parameters are
    attributes (class-level ``default_params`` overridden by keyword arguments), the model is
    the three laws ``PX0`` / ``PX`` / ``PY`` (+ ``proposal0`` / ``proposal`` / ``logeta`` for the
    guided and auxiliary filters), each returning a ``ProbDist``."""

    default_params = {}

    def __init__(self, **kwargs):
        settings = dict(type(self).default_params)
        settings.update(kwargs)
        vars(self).update(settings)

    def _undefined(self, what):
        return NotImplementedError("%s does not define %s()" % (type(self).__name__, what))

    def PX0(self):
        """Law of X_0."""
        raise self._undefined("PX0")

    def PX(self, t, xp):
        """Law of X_t given X_{t-1} = xp."""
        raise self._undefined("PX")

    def PY(self, t, xp, x):
        """Law of Y_t given X_{t-1} = xp and X_t = x."""
        raise self._undefined("PY")

    def proposal0(self, data):
        raise self._undefined("proposal0")

    def proposal(self, t, xp, data):
        raise self._undefined("proposal")

    def simulate_given_x(self, x):
        """One observation per state of the path x (a list); the state before the first is None."""
        previous = [None] + list(x[:-1])
        return [self.PY(t, x_before, x_now).rvs(size=1)
                for t, (x_before, x_now) in enumerate(zip(previous, x))]

    def simulate(self, T):
        """Simulate state and observation processes (state_space_models.py:278-296):
        lists x, y of length T -- the whole state path first, then the observations, which is
        the order the reference consumes numpy's stream in."""
        path = []
        while len(path) < T:
            law = self.PX(len(path), path[-1]) if path else self.PX0()
            path.append(law.rvs(size=1))
        return path, self.simulate_given_x(path)

    def _device_params(self, fk_kind):
        """(model kind, dx, dy, params (16,) or None, matrices) for the fused
        step loop, or None when the model is outside the closed family."""
        return None


class Bootstrap(FeynmanKac):
    """Bootstrap Feynman-Kac formalism of a state-space model (state_space_models.py:299-349):
    particles move with the model's own transition -- ``_kernel`` -- and are weighted by the
    observation density."""

    _fk_kind = _lib.FK_BOOTSTRAP

    def __init__(self, ssm=None, data=None):
        self.ssm, self.data = ssm, data
        self.du = ssm.PX0().dim

    @property
    def T(self):
        return len(self.data) if self.data is not None else 0

    def _kernel(self, t, xp):
        """The law the particles of step t are drawn from (xp None: step 0)."""
        return self.ssm.PX0() if xp is None else self.ssm.PX(t, xp)

    def M0(self, N):
        return self._kernel(0, None).rvs(size=N)

    def M(self, t, xp):
        return self._kernel(t, xp).rvs(size=len(xp))

    def Gamma0(self, u):                                   # state_space_models.py:335-340: the SQMC maps
        return self._kernel(0, None).ppf(u)

    def Gamma(self, t, xp, u):
        return self._kernel(t, xp).ppf(u)

    def logG(self, t, xp, x):
        return self.ssm.PY(t, xp, x).logpdf(self.data[t])

    def logpt(self, t, xp, x):
        """log-density of X_t = x given X_{t-1} = xp."""
        return self.ssm.PX(t, xp).logpdf(x)

    def _device_model(self):
        """Parameters of the fused device loop, or None.  The fused kernels implement the STOCK
        model: it is chosen only when neither this Feynman-Kac class nor the state-space model
        overrides one of the methods the kernels stand for -- a user subclass that redefines
        ``PY`` (or ``logG``, ``time_to_resample``, ...) must run its own code, through the
        template-method path (the reference's normal way to customise a model)."""
        base = GuidedPF if isinstance(self, GuidedPF) else Bootstrap
        cls = type(self)
        for name in ("M0", "M", "logG", "_kernel", "time_to_resample", "done"):
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
    (state_space_models.py:352-398): particles move with the proposal and carry the
    importance ratio transition x observation / proposal."""

    _fk_kind = _lib.FK_GUIDED

    def _kernel(self, t, xp):
        return self.ssm.proposal0(self.data) if xp is None else self.ssm.proposal(t, xp, self.data)

    def logG(self, t, xp, x):
        start = t == 0
        moved = self.ssm.PX0() if start else self.ssm.PX(t, xp)
        ratio = moved.logpdf(x) + self.ssm.PY(t, xp, x).logpdf(self.data[t])
        return ratio - self._kernel(t, None if start else xp).logpdf(x)


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
    """APF whose proposal is the transition kernel (state_space_models.py:431-438): the bootstrap move and
    weight, resampling on ``lw + logeta``.  Fused for the stock ``StochVol`` and ``LinearGauss``
    (``SMC_FK_APF_BOOT``: the bootstrap step of the kernels with the auxiliary weights of ``AuxiliaryPF``)."""

    _fk_kind = _lib.FK_APF_BOOT

    def _device_model(self):
        base = Bootstrap._device_model(self)
        if base is None or not base.get("apf") or type(self).logeta is not APFMixin.logeta:
            return None
        if base["kind"] not in (_lib.MODEL_STOCHVOL, _lib.MODEL_LINGAUSS):
            return None                                    # (the multivariate auxiliary filter is the guided one)
        owner = next((c for c in type(self.ssm).__mro__ if "_device_params" in vars(c)), None)
        if getattr(type(self.ssm), "logeta", None) is not getattr(owner, "logeta", None):
            return None
        return base


class _AR1State(StateSpaceModel):
    """Shared by the models whose state is a Gaussian AR(1) around a level: the level / persistence /
    innovation sd are named by ``_ar1 = (level, persistence, sd)`` (attribute names)."""
    _ar1 = ("mu", "rho", "sigma")

    def _ar1_values(self):
        return tuple(getattr(self, name) for name in self._ar1)

    def sig0(self):
        """sd of the stationary law."""
        _, persistence, sd = self._ar1_values()
        return sd / np.sqrt(1.0 - persistence ** 2)


class StochVol(_AR1State):
    r"""Univariate stochastic volatility model (state_space_models.py:446-473).

    X_0 ~ N(mu, sigma^2/(1-rho^2)); X_t = mu + rho (X_{t-1}-mu) + sigma U_t;
    Y_t | X_t ~ N(0, exp(X_t)).
    """
    default_params = {"mu": -1.02, "rho": 0.9702, "sigma": 0.178}

    def EXt(self, xp):
        """E[X_t | X_{t-1} = xp]"""
        drift = (1.0 - self.rho) * self.mu
        return drift + self.rho * xp

    def PX0(self):
        return _gauss(self.mu, self.sig0())

    def PX(self, t, xp):
        return _gauss(self.EXt(xp), self.sigma)

    def PY(self, t, xp, x):
        return _gauss(0.0, np.exp(0.5 * x))

    # ---- Pitt & Shephard's proposal and auxiliary function (state_space_models.py:475-498): a Gaussian
    # centred at one Newton-like step from the predictive mean towards the mode of the observation term
    def _xhat(self, centre, sd, y):
        return centre + 0.5 * sd ** 2 * (y ** 2 * np.exp(-centre) - 1.0)

    def proposal0(self, data):
        sd = self.sig0()
        return _gauss(self._xhat(0.0, sd, data[0]), sd)

    def proposal(self, t, xp, data):
        return _gauss(self._xhat(self.EXt(xp), self.sigma, data[t]), self.sigma)

    def logeta(self, t, x, data):
        y_next = data[t + 1]
        ahead = self.EXt(x)
        d_ahead = ahead - self.mu
        d_hat = self._xhat(ahead, self.sigma, y_next) - self.mu
        quad = 0.5 / self.sigma ** 2 * (d_hat ** 2 - d_ahead ** 2)
        return quad - 0.5 * y_next ** 2 * np.exp(-ahead) * (1.0 + d_ahead)

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
        # the standardised innovation of the state, then the conditional law of a correlated Gaussian
        z = (x - self.mu) / self.sig0() if t == 0 else (x - self.EXt(xp)) / self.sigma
        vol = np.exp(0.5 * x)
        return _gauss(vol * self.phi * z, vol * np.sqrt(1.0 - self.phi ** 2))

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

    def _forcing(self, t):
        """The time-dependent term of the transition (host cos, as the device's aux tape holds it)."""
        return self.d * np.cos(self.e * (t - 1))

    def PX0(self):
        return _gauss(scale=2.0)

    def PX(self, t, xp):
        pulled = self.b * xp + self.c * xp / (1.0 + xp ** 2)
        return _gauss(pulled + self._forcing(t), self.sigmaX)

    def PY(self, t, xp, x):
        return _gauss(self.a * x ** 2)

    def _device_params(self, fk_kind):
        if fk_kind != _lib.FK_BOOTSTRAP:
            return None
        p = np.zeros(_lib.PARAM_STRIDE)
        p[[0, 1, 2, 3, 5]] = [self.b, self.sigmaX, self.c, 2.0, self.a]
        return dict(kind=_lib.MODEL_GORDON, dx=1, dy=1, params=p, aux=lambda T: self._forcing(np.arange(T)))


class BearingsOnly(StateSpaceModel):
    """Bearings-only tracking (state_space_models.py:580-606): positions with Gaussian increments, velocities carried by
    Dirac laws, the bearing observed.  No fused descriptor: it runs on the template-method step (device operators for
    the Normal components, the weights and the resampling)."""
    default_params = {"sigmaX": 2.0e-4, "sigmaY": 1e-3, "x0": np.array([3e-3, -3e-3, 1.0, 1.0])}

    def PX0(self):
        return dists.IndepProd(dists.Normal(loc=self.x0[0], scale=self.sigmaX), dists.Normal(loc=self.x0[1], scale=self.sigmaX),
                               dists.Dirac(loc=self.x0[2]), dists.Dirac(loc=self.x0[3]))

    def PX(self, t, xp):
        xp = np.asarray(xp)
        return dists.IndepProd(dists.Normal(loc=xp[:, 0], scale=self.sigmaX), dists.Normal(loc=xp[:, 1], scale=self.sigmaX),
                               dists.Dirac(loc=xp[:, 0] + xp[:, 2]), dists.Dirac(loc=xp[:, 1] + xp[:, 3]))

    def PY(self, t, xp, x):
        x = np.asarray(x)
        angle = np.arctan(x[:, 3] / x[:, 2])
        angle[x[:, 2] < 0.0] += np.pi
        return dists.Normal(loc=angle, scale=self.sigmaY)


class MVStochVol(StateSpaceModel):
    """Multivariate stochastic volatility (state_space_models.py:630-655): X_t - mu = F (X_{t-1} - mu) + U_t, U_t ~ N(0, covX);
    Y_t(k) = exp(X_t(k) / 2) V_t(k), V_t ~ N(0, corY).  No fused descriptor (the observation law's scale varies by particle
    AND component): the template-method step, MvNormal on the device operators."""
    default_params = {"mu": 0.0, "covX": None, "corY": None, "F": None}

    def offset(self):
        return self.mu - np.dot(self.F, self.mu)

    def PX0(self):
        return dists.MvNormal(loc=self.mu, cov=self.covX)

    def PX(self, t, xp):
        return dists.MvNormal(loc=np.dot(np.asarray(xp), self.F.T) + self.offset(), cov=self.covX)

    def PY(self, t, xp, x):
        return dists.MvNormal(loc=np.zeros(np.shape(x)[-1]), scale=np.exp(0.5 * np.asarray(x)), cov=self.corY)


class DiscreteCox(_AR1State):
    r"""A discrete Cox model (state_space_models.py:611-630).

    Y_t | X_t = x ~ Poisson(e^x);  X_t = mu + phi (X_{t-1} - mu) + U_t, U_t ~ N(0, sigma^2);
    X_0 ~ N(mu, sigma^2 / (1 - phi^2)).
    """
    default_params = {"mu": 0.0, "sigma": 1.0, "phi": 0.95}
    _ar1 = ("mu", "phi", "sigma")

    def PX0(self):
        return _gauss(self.mu, self.sig0())

    def PX(self, t, xp):
        return _gauss(self.mu + self.phi * (xp - self.mu), self.sigma)

    def PY(self, t, xp, x):
        return dists.Poisson(rate=np.exp(x))

    def _device_params(self, fk_kind):
        if fk_kind != _lib.FK_BOOTSTRAP:
            return None
        p = np.zeros(_lib.PARAM_STRIDE)
        p[:4] = [self.mu, self.phi, self.sigma, self.sig0()]

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
        return _gauss()

    def PX(self, t, xp):
        grown = xp + self.tau0                          # (the reference's association: (x + tau0) - tau1 exp(.))
        return _gauss(grown - self.tau1 * np.exp(self.tau2 * xp), self.sigmaX)

    def PY(self, t, xp, x):
        return _gauss(x, self.sigmaY)

    def _device_params(self, fk_kind):
        if fk_kind != _lib.FK_BOOTSTRAP:
            return None
        p = np.zeros(_lib.PARAM_STRIDE)
        p[:7] = [self.tau0, self.sigmaX, self.sigmaY, 1.0, np.log(self.sigmaY), self.tau1,
                 self.tau2]
        return dict(kind=_lib.MODEL_THETALOGISTIC, dx=1, dy=1, params=p)
