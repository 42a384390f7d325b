"""Phase timeline of the strict step's two launches (k_strict_classify incl. the island's chain, k_strict_search) from a
-DSMC_TRACE build (perf diagnostics; build: ABLS=TRACE bash tools/build_ablations.sh, run on the GPU box).
    python tools/trace_strict.py [log2N] [scheme] [sv]
Stamps are wall_clock64() ticks (100 MHz) by thread 0 of every workgroup of island 0, printed in microseconds relative
to the kernel's first stamp: min / median / p99 / max over workgroups."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SMC_HIP_LIBRARY"] = os.path.join(ROOT, "particles_amd", "lib", "abl", "libsmc_TRACE.so")
sys.path.insert(0, ROOT)
import particles_amd as pa                                      # noqa: E402
from particles_amd import _lib, kalman, state_space_models as ssm   # noqa: E402
from bench import synthetic_data                                # noqa: E402

log2N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
scheme = sys.argv[2] if len(sys.argv) > 2 else "systematic"
N = 1 << log2N
y = synthetic_data(200)
model = ssm.StochVol() if len(sys.argv) > 3 and sys.argv[3] == "sv" else kalman.ToySSM(0.2)
pf = pa.SMC(fk=ssm.Bootstrap(ssm=model, data=y), N=N, seed=123, use_graph=False, resampling=scheme,
            ESSrmin=1.0 if len(sys.argv) > 3 else 0.5, strict_ancestors=True)
pf.step_async(100)
pf.sync()
nt = N // 1024
buf = np.zeros((2 * nt + 8) * 8, dtype=np.uint64)
lib = _lib.lib()
lib.smc_debug_trace_strict.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
_lib.check(lib.smc_debug_trace_strict(pf._f, buf.ctypes.data_as(ctypes.c_void_p)))
st = buf.reshape(2 * nt + 8, 8)


def show(name, rows, labels, t0=None):
    rows = rows[rows[:, 0] > 0]
    t0 = rows[:, 0].min() if t0 is None else t0
    print("%s: %d workgroups" % (name, rows.shape[0]))
    for k, lab in enumerate(labels):
        col = rows[:, k].astype(np.int64)
        col = col[col > 0] - int(t0)
        if col.size:
            print("  %-26s n=%5d  min %6.2f  median %6.2f  p99 %6.2f  max %6.2f us"
                  % (lab, col.size, col.min() / 100.0, np.median(col) / 100.0, np.percentile(col, 99) / 100.0, col.max() / 100.0))
    return t0


rows = st[:nt]
late = np.argsort(-rows[:, 5].astype(np.int64))[:8]
t00 = rows[rows[:, 0] > 0][:, 0].min()
print("latest tickets (workgroup: start, reduced, ticket us):",
      [(int(i), round((int(rows[i, 0]) - int(t00)) / 100.0, 2), round((int(rows[i, 1]) - int(t00)) / 100.0, 2),
        round((int(rows[i, 5]) - int(t00)) / 100.0, 2)) for i in late])
t0 = show("k_strict_classify", st[:nt], ["start", "reduced: K, s, before", "classified", "prefixes", "published", "ticket taken"])
show("  the island's chain (last workgroup)", st[nt:nt + 1], ["start", "loads back", "tile prefixes", "sorted", "walked", "end"], t0)
desc = ctypes.create_string_buffer(256)
_lib.check(lib.smc_filter_describe(pf._f, desc, 256))
one = b"k_strict_step" in desc.value
print("(%s)" % desc.value.decode())
show("k_strict_search" + (" (same launch: times from the launch's first stamp)" if one else ""), st[nt + 8:], ["start", "record + su", "tile staged", "range known", "end", "last pass: thresholds", "last pass: bisected"], t0 if one else None)
