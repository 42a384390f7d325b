"""Linear Gaussian state-space models of ``particles.kalman`` that are on the
hot path: ``MVLinearGauss`` (:296-361), ``MVLinearGauss_Guarniero_etal``
(:364-394) and ``LinearGauss`` (:397-452), including the optimal proposals the
guided filter uses.  (The exact ``Kalman`` filter/smoother of the reference is
a CPU-sized algorithm and is not re-implemented; the test oracle restates it.)
"""
import numpy as np

from . import _lib
from . import distributions as dists
from . import state_space_models as ssms

error_msg = "arguments of KalmanFilter.__init__ have inconsistent shapes"


class MVLinearGauss(ssms.StateSpaceModel):
    r"""Multivariate linear Gaussian model (kalman.py:296-361).

    X_0 ~ N(mu0, cov0);  X_t = F X_{t-1} + U_t, U_t ~ N(0, covX);
    Y_t = G X_t + V_t, V_t ~ N(0, covY).
    """

    def __init__(self, F=None, G=None, covX=None, covY=None, mu0=None, cov0=None):
        self.covX, self.covY = np.atleast_2d(covX), np.atleast_2d(covY)
        self.dx, self.dy = self.covX.shape[0], self.covY.shape[0]
        self.mu0 = np.zeros(self.dx) if mu0 is None else mu0
        self.cov0 = self.covX if cov0 is None else np.atleast_2d(cov0)
        self.F = np.eye(self.dx) if F is None else np.atleast_2d(F)
        self.G = np.eye(self.dy, self.dx) if G is None else np.atleast_2d(G)
        self.check_shapes()

    def check_shapes(self):
        assert self.covX.shape == (self.dx, self.dx), error_msg
        assert self.covY.shape == (self.dy, self.dy), error_msg
        assert self.F.shape == (self.dx, self.dx), error_msg
        assert self.G.shape == (self.dy, self.dx), error_msg
        assert self.mu0.shape == (self.dx,), error_msg
        assert self.cov0.shape == (self.dx, self.dx), error_msg

    def PX0(self):
        return dists.MvNormal(loc=self.mu0, cov=self.cov0)

    def PX(self, t, xp):
        return dists.MvNormal(loc=np.dot(xp, self.F.T), cov=self.covX)

    def PY(self, t, xp, x):
        return dists.MvNormal(loc=np.dot(x, self.G.T), cov=self.covY)

    def _gain(self, pred_cov):
        """Kalman gain and filtered covariance for a predictive covariance
        (kalman.py:215-229): shared by all particles."""
        S = self.G @ pred_cov @ self.G.T + self.covY
        K = np.linalg.solve(S, (pred_cov @ self.G.T).T).T
        return K, pred_cov - K @ self.G @ pred_cov

    def proposal(self, t, xp, data):
        """N(m + K (y_t - G m), covX - K G covX), m = F xp  (kalman.py:348-351)."""
        m = np.dot(xp, self.F.T)
        K, fc = self._gain(self.covX)
        return dists.MvNormal(loc=m + np.dot(data[t] - np.dot(m, self.G.T), K.T), cov=fc)

    def proposal0(self, data):
        K, fc = self._gain(self.cov0)
        return dists.MvNormal(loc=self.mu0 + np.dot(data[0] - np.dot(self.mu0, self.G.T), K.T),
                              cov=fc)

    def logeta(self, t, x, data):
        """log p(y_{t+1} | x_t = x) = log N(y_{t+1}; G F x, G covX G' + covY)  (kalman.py:358-361: the
        ``logpyt`` of ``filter_step`` on the one-step prediction from x)."""
        S = self.G @ self.covX @ self.G.T + self.covY
        return dists.MvNormal(loc=np.dot(np.dot(x, self.F.T), self.G.T), cov=S).logpdf(data[t + 1])

    def _device_params(self, fk_kind):
        if not (1 <= self.dy <= self.dx <= 32):      # k_propagate_mv's range; beyond: generic path
            return None
        return dict(kind=_lib.MODEL_MVLINGAUSS, dx=self.dx, dy=self.dy, params=None,
                    F=self.F, G=self.G, covX=self.covX, covY=self.covY,
                    mu0=self.mu0, cov0=self.cov0, apf=True)


class MVLinearGauss_Guarniero_etal(MVLinearGauss):
    """G = covX = covY = cov0 = I, F[i,j] = alpha^(1+|i-j|)  (kalman.py:364-394)."""

    def __init__(self, alpha=0.4, dx=2):
        F = np.empty((dx, dx))
        for i in range(dx):
            for j in range(dx):
                F[i, j] = alpha ** (1 + abs(i - j))
        MVLinearGauss.__init__(self, F=F, G=np.eye(dx), covX=np.eye(dx), covY=np.eye(dx))


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
        if self.sigma0 is None:
            self.sigma0 = self.sigmaX / np.sqrt(1.0 - self.rho ** 2)
        MVLinearGauss.__init__(self, F=self.rho, G=1.0, covX=self.sigmaX ** 2,
                               covY=self.sigmaY ** 2, cov0=self.sigma0 ** 2)

    def PX0(self):
        return dists.Normal(scale=self.sigma0)

    def PX(self, t, xp):
        return dists.Normal(loc=self.rho * xp, scale=self.sigmaX)

    def PY(self, t, xp, x):
        return dists.Normal(loc=x, scale=self.sigmaY)

    def proposal0(self, data):
        sig2post = 1.0 / (1.0 / self.sigma0 ** 2 + 1.0 / self.sigmaY ** 2)
        mupost = sig2post * (data[0] / self.sigmaY ** 2)
        return dists.Normal(loc=mupost, scale=np.sqrt(sig2post))

    def proposal(self, t, xp, data):
        sig2post = 1.0 / (1.0 / self.sigmaX ** 2 + 1.0 / self.sigmaY ** 2)
        mupost = sig2post * (self.rho * xp / self.sigmaX ** 2 + data[t] / self.sigmaY ** 2)
        return dists.Normal(loc=mupost, scale=np.sqrt(sig2post))

    def logeta(self, t, x, data):
        """Auxiliary function of the APF (kalman.py:448-452): the predictive density of y_{t+1}."""
        law = dists.Normal(loc=self.rho * x, scale=np.sqrt(self.sigmaX ** 2 + self.sigmaY ** 2))
        return law.logpdf(data[t + 1])

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
