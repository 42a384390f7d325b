"""Phase timeline of the one-pass uniform_spacings kernel (multinomial, Philox draws) of the last step, from a
-DSMC_TRACE build (tools/trace_step.py's conventions): workgroup 0 is the island's reduction (k_reduce2's function),
the others draw, wait for its decision, publish, look back and write."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SMC_HIP_LIBRARY"] = os.path.join(ROOT, "particles_amd", "lib", "abl", "libsmc_TRACE.so")
sys.path.insert(0, ROOT)
import particles_amd as pa                                      # noqa: E402
from particles_amd import _lib, state_space_models as ssm   # noqa: E402

log2N = int(sys.argv[1]) if len(sys.argv) > 1 else 22
N = 1 << log2N
rng = np.random.RandomState(42)
y = [np.array([v]) for v in rng.standard_normal(120)]
pf = pa.SMC(fk=ssm.Bootstrap(ssm=ssm.StochVol(), data=y), N=N, seed=123, resampling="multinomial", ESSrmin=1.0)
pf.step_async(60)
pf.sync()
ntiles = N // 1024
buf = np.zeros((2 * ntiles + 8) * 8, dtype=np.uint64)
lib = _lib.lib()
lib.smc_debug_trace_strict.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
_lib.check(lib.smc_debug_trace_strict(pf._f, buf.ctypes.data_as(ctypes.c_void_p)))
st = buf.reshape(-1, 8)
st = st[st[:, 0] > 0]
t0 = int(st[:, 0].min())
print("k_f_spacing_onepass at N = 2^%d: %d workgroups stamped (workgroup 0 = the island's reduction)" % (log2N, st.shape[0]))
r = st[0]
print("  workgroup 0 (reduction): start %.2f  done (decision published on the way, shares written) %.2f us"
      % ((int(r[0]) - t0) / 100.0, (int(r[5]) - t0) / 100.0))
for k, lab in enumerate(["start", "tables staged, t known", "draws done, offsets stored", "look-back done", "decision seen", "tile prefixes written"]):
    col = st[1:, k].astype(np.int64)
    col = col[col >= t0] - t0
    if col.size:
        print("  %-24s n=%5d  min %6.2f  median %6.2f  p90 %6.2f  max %6.2f us"
              % (lab, col.size, col.min() / 100.0, np.median(col) / 100.0, np.percentile(col, 90) / 100.0, col.max() / 100.0))
