"""Phase timeline of k_move from a -DSMC_TRACE build (tools/build_trace.sh)."""
import ctypes, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SMC_HIP_LIBRARY"] = os.path.join(ROOT, "particles_amd", "lib", "abl", "libsmc_TRACE.so")
import particles_amd as pa
from particles_amd import _lib, kalman, state_space_models as ssm
ess = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
N = 1 << 20
rng = np.random.RandomState(42)
x = np.cumsum(rng.standard_normal(80)); y = [np.array([v]) for v in x + 0.2 * rng.standard_normal(80)]
pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=N, collect="off", seed=1, ESSrmin=ess, use_graph=False)
pf.step_async(60); pf.sync()
L = _lib.lib()
nt = N // 512
buf = np.zeros((nt, 8), dtype=np.uint64)
L.smc_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
L.smc_debug_trace(pf._f, buf.ctypes.data_as(ctypes.c_void_p))
t = buf.astype(np.float64)
t0 = t[:, 0].min()
names = ["entry", "info", "loads+RNG", "compute+stores", "lse_block", "publish", "ticket", "finalize(last)"]
print("ESSrmin", ess, " wall_clock64 ticks (100 MHz => 10 ns each)")
for k in range(8):
    col = t[:, k]; col = col[col > 0] - t0
    if col.size:
        print("%-16s n=%4d  min %7.0f  med %7.0f  p90 %7.0f  max %7.0f ns" % (names[k], col.size, col.min() * 10, np.median(col) * 10, np.percentile(col, 90) * 10, col.max() * 10))
d = (t[:, 6] - t[:, 0]) * 10
print("block lifetime: med %.0f  max %.0f ns; spread of entry: %.0f ns" % (np.median(d), d.max(), (t[:, 0].max() - t0) * 10))
