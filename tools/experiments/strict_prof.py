import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, particles_amd as pa
from particles_amd import kalman, state_space_models as ssm
y = bench.synthetic_data(80)
pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=1 << 20, seed=5, collect="off", strict_ancestors=True)
pf.step_async(80); pf.sync()
