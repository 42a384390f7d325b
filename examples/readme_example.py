"""The example of the reference's README (README.md:60-90) on the MI355X path: only the
import lines differ.  Run on a GPU box:  python examples/readme_example.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import particles_amd as particles                       # import particles
from particles_amd import distributions as dists        # from particles import distributions as dists
from particles_amd import state_space_models as ssm     # from particles import state_space_models as ssm
from particles_amd.collectors import Moments            # from particles.collectors import Moments


class ToySSM(ssm.StateSpaceModel):
    def PX0(self):                    # Distribution of X_0
        return dists.Normal()         # X_0 ~ N(0, 1)

    def PX(self, t, xp):              # Distribution of X_t given X_{t-1}
        return dists.Normal(loc=xp)   # X_t ~ N(X_{t-1}, 1)

    def PY(self, t, xp, x):           # Distribution of Y_t given X_t (and X_{t-1})
        return dists.Normal(loc=x, scale=self.sigma)   # Y_t ~ N(X_t, sigma^2)


np.random.seed(42)
my_model = ToySSM(sigma=0.2)
x, y = my_model.simulate(200)         # sample size is 200

# a user-defined model runs the template-method step with device operators ...
alg = particles.SMC(fk=ssm.Bootstrap(ssm=my_model, data=y), N=200, collect=[Moments()])
alg.run()
print("user-defined model  : logLt = %.4f, filtering mean at T-1 = %.4f"
      % (alg.logLt, alg.summaries.moments[-1]["mean"]))

# ... with its particles resident in HBM and the device generator, at N = 2^20
from particles_amd import resampling as rs              # noqa: E402
particles.set_resident(True)
rs.set_rng("philox")
res = particles.SMC(fk=ssm.Bootstrap(ssm=my_model, data=y), N=1 << 20, collect="off")
res.run()
print("same model, resident : logLt = %.4f in %.1f ms (N = 2^20)" % (res.logLt, 1e3 * res.cpu_time))
particles.set_resident(False)
rs.set_rng("numpy")

# ... a model of the fused family (here the same one, spelled as LinearGauss) runs the whole
# T-loop on the device, here with 2^20 particles and the complete history kept in HBM
from particles_amd import kalman                        # noqa: E402
big = particles.SMC(fk=ssm.Bootstrap(ssm=kalman.LinearGauss(rho=1.0, sigmaX=1.0, sigmaY=0.2, sigma0=1.0),
                                     data=y), N=1 << 20, store_history=True, seed=1)
big.run()
traj = big.hist.compute_trajectories()
print("fused loop, N = 2^20 : logLt = %.4f in %.1f ms; %d distinct ancestors at t = 0"
      % (big.logLt, 1e3 * big.cpu_time, np.unique(traj[0]).size))
