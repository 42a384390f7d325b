"""Linear Gaussian state-space models of ``particles.kalman`` that are on the
hot path: ``MVLinearGauss`` (:296-361), ``MVLinearGauss_Guarniero_etal``
(:364-394) and ``LinearGauss`` (:397-452), including the optimal proposals the
guided filter uses.  (The exact ``Kalman`` filter/smoother of the reference is
a CPU-sized algorithm and is not re-implemented; the test oracle restates it.)

Same class names, constructor arguments, attributes and laws as the reference; the
bodies are this package's own: a linear Gaussian model is held as its six arrays, every
law it hands out is one conditioning step of ``_condition`` (multivariate) or
``_scalar_posterior`` (univariate), and the arithmetic of those steps follows the
reference's operation for operation (the golden fixtures of tests/golden pin the bits).
"""
import numpy as np

from . import _lib
from . import distributions as dists
from . import state_space_models as ssms


def _as_matrix(value, fallback):
    return fallback if value is None else np.atleast_2d(value)


def _mv(mean, cov):
    return dists.MvNormal(loc=mean, cov=cov)


_gauss = ssms._gauss


class MVLinearGauss(ssms.StateSpaceModel):
    r"""Multivariate linear Gaussian model (kalman.py:296-361).

    X_0 ~ N(mu0, cov0);  X_t = F X_{t-1} + U_t, U_t ~ N(0, covX);
    Y_t = G X_t + V_t, V_t ~ N(0, covY).  Only covX and covY are mandatory
    (they fix dx and dy); mu0 = 0, cov0 = covX, F = I, G = [I 0] otherwise.
    """
    _SHAPES = (("covX", "dx", "dx"), ("covY", "dy", "dy"), ("F", "dx", "dx"), ("G", "dy", "dx"),
               ("mu0", "dx", None), ("cov0", "dx", "dx"))

    def __init__(self, F=None, G=None, covX=None, covY=None, mu0=None, cov0=None):
        self.covX = np.atleast_2d(covX)
        self.covY = np.atleast_2d(covY)
        self.dx = self.covX.shape[0]
        self.dy = self.covY.shape[0]
        self.F = _as_matrix(F, np.eye(self.dx))
        self.G = _as_matrix(G, np.eye(self.dy, self.dx))
        self.cov0 = _as_matrix(cov0, self.covX)
        self.mu0 = mu0 if mu0 is not None else np.zeros(self.dx)
        self.check_shapes()

    def check_shapes(self):
        """AssertionError (as the reference raises) naming the first array whose shape does not fit."""
        for name, rows, cols in self._SHAPES:
            want = (getattr(self, rows),) + (() if cols is None else (getattr(self, cols),))
            got = np.shape(getattr(self, name))
            assert got == want, "MVLinearGauss: %s has shape %s, the model's dimensions need %s" % (name, got, want)

    # ---- the three laws that define the model
    def PX0(self):
        return _mv(self.mu0, self.cov0)

    def PX(self, t, xp):
        return _mv(self._push(xp), self.covX)

    def PY(self, t, xp, x):
        return _mv(np.dot(x, self.G.T), self.covY)

    def _push(self, x):
        """E[X_t | X_{t-1} = x], row-wise."""
        return np.dot(x, self.F.T)

    def _gain(self, pred_cov):
        """Kalman gain and filtered covariance for a predictive covariance
        (kalman.py:215-229): shared by all particles."""
        S = self.G @ pred_cov @ self.G.T + self.covY
        K = np.linalg.solve(S, (pred_cov @ self.G.T).T).T
        return K, pred_cov - K @ self.G @ pred_cov

    def _condition(self, mean, cov, y):
        """N(mean, cov) conditioned on the observation y of it: one filtering step, for a
        batch of means with one covariance."""
        K, post_cov = self._gain(cov)
        innovation = y - np.dot(mean, self.G.T)
        return _mv(mean + np.dot(innovation, K.T), post_cov)

    # ---- the optimal proposals of the guided filter and the APF's auxiliary function
    def proposal0(self, data):
        return self._condition(self.mu0, self.cov0, data[0])

    def proposal(self, t, xp, data):
        """N(m + K (y_t - G m), covX - K G covX), m = F xp  (kalman.py:348-351)."""
        return self._condition(self._push(xp), self.covX, data[t])

    def logeta(self, t, x, data):
        """log p(y_{t+1} | x_t = x) = log N(y_{t+1}; G F x, G covX G' + covY)  (kalman.py:358-361: the
        ``logpyt`` of ``filter_step`` on the one-step prediction from x)."""
        S = self.G @ self.covX @ self.G.T + self.covY
        return _mv(np.dot(self._push(x), self.G.T), S).logpdf(data[t + 1])

    def _device_params(self, fk_kind):
        if not (1 <= self.dy <= self.dx <= 32):      # k_propagate_mv's range; beyond: generic path
            return None
        return dict(kind=_lib.MODEL_MVLINGAUSS, dx=self.dx, dy=self.dy, params=None,
                    F=self.F, G=self.G, covX=self.covX, covY=self.covY,
                    mu0=self.mu0, cov0=self.cov0, apf=True)


class MVLinearGauss_Guarniero_etal(MVLinearGauss):
    """G = covX = covY = cov0 = I, F[i,j] = alpha^(1+|i-j|)  (kalman.py:364-394)."""

    def __init__(self, alpha=0.4, dx=2):
        # (Python's scalar power, entry by entry: the values the reference's double loop produces)
        band = [[alpha ** (1 + abs(row - col)) for col in range(dx)] for row in range(dx)]
        identity = np.eye(dx)
        super().__init__(F=np.array(band), G=identity, covX=identity, covY=np.eye(dx))


class LinearGauss(MVLinearGauss):
    r"""A basic (univariate) linear Gaussian model (kalman.py:397-452).

    X_0 ~ N(0, sigma0^2); X_t | X_{t-1} ~ N(rho X_{t-1}, sigmaX^2);
    Y_t | X_t ~ N(X_t, sigmaY^2).  sigma0=None -> the stationary value.
    The README's ``ToySSM(sigma)`` is ``LinearGauss(rho=1, sigmaX=1, sigmaY=sigma,
    sigma0=1)``.
    """
    default_params = {"sigmaY": 0.2, "rho": 0.9, "sigmaX": 1.0, "sigma0": None}

    def __init__(self, **kwargs):
        ssms.StateSpaceModel.__init__(self, **kwargs)
        if self.sigma0 is None:                      # the stationary law of the AR(1) state
            self.sigma0 = self.sigmaX / np.sqrt(1.0 - self.rho ** 2)
        variances = {"covX": self.sigmaX ** 2, "covY": self.sigmaY ** 2, "cov0": self.sigma0 ** 2}
        MVLinearGauss.__init__(self, F=self.rho, G=1.0, **variances)

    def PX0(self):
        return _gauss(scale=self.sigma0)

    def PX(self, t, xp):
        return _gauss(self.rho * xp, self.sigmaX)

    def PY(self, t, xp, x):
        return _gauss(x, self.sigmaY)

    def _scalar_posterior(self, prior_sd, y, prior_mean=None):
        """N(prior_mean, prior_sd^2) (mean 0 when None) given y = x + N(0, sigmaY^2): precisions add,
        the mean is the precision-weighted one -- written with the reference's divisions
        (kalman.py:436-446) so that the bits agree."""
        var_post = 1.0 / (1.0 / prior_sd ** 2 + 1.0 / self.sigmaY ** 2)
        pull = y / self.sigmaY ** 2
        if prior_mean is not None:
            pull = prior_mean / prior_sd ** 2 + pull
        return _gauss(var_post * pull, np.sqrt(var_post))

    def proposal0(self, data):
        return self._scalar_posterior(self.sigma0, data[0])

    def proposal(self, t, xp, data):
        return self._scalar_posterior(self.sigmaX, data[t], prior_mean=self.rho * xp)

    def logeta(self, t, x, data):
        """Auxiliary function of the APF (kalman.py:448-452): the predictive density of y_{t+1}."""
        spread = np.sqrt(self.sigmaX ** 2 + self.sigmaY ** 2)
        return _gauss(self.rho * x, spread).logpdf(data[t + 1])

    def _device_params(self, fk_kind):
        p = np.zeros(_lib.PARAM_STRIDE)
        sx2, sy2 = self.sigmaX ** 2, self.sigmaY ** 2
        s2p = 1.0 / (1.0 / sx2 + 1.0 / sy2)
        s2p0 = 1.0 / (1.0 / self.sigma0 ** 2 + 1.0 / sy2)
        p[:15] = [self.rho, self.sigmaX, self.sigmaY, self.sigma0,
                  np.log(self.sigmaY), np.log(self.sigmaX), np.log(self.sigma0), sx2, sy2,
                  s2p, np.sqrt(s2p), np.log(np.sqrt(s2p)),
                  s2p0, np.sqrt(s2p0), np.log(np.sqrt(s2p0))]
        p[15] = np.sqrt(sx2 + sy2)                 # scale of logeta (APF)
        return dict(kind=_lib.MODEL_LINGAUSS, dx=1, dy=1, params=p, apf=True)


def ToySSM(sigma=0.2):
    """The README's example model (README.md:66-72) as a ``LinearGauss``."""
    return LinearGauss(rho=1.0, sigmaX=1.0, sigmaY=sigma, sigma0=1.0)
