"""Phase timeline of the last step's kernels from a -DSMC_TRACE build
(perf diagnostics; build: ABLS=TRACE bash tools/build_ablations.sh, run on the GPU box).

Stamps are wall_clock64() ticks (100 MHz) taken by thread 0 of every workgroup;
printed relative to the kernel's first stamp, in microseconds: min / median / max
over workgroups."""
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
y = synthetic_data(1400)
model = ssm.StochVol() if len(sys.argv) > 3 and sys.argv[3] == "sv" else kalman.ToySSM(0.2)
pf = pa.SMC(fk=ssm.Bootstrap(ssm=model, data=y), N=N, seed=123, use_graph=False, resampling=scheme,
            ESSrmin=1.0 if len(sys.argv) > 2 else 0.5)
pf.step_async(100)
pf.sync()

nparts, ntiles = N // 1024, N // 1024
buf = np.zeros((nparts + ntiles) * 8, dtype=np.uint64)
lib = _lib.lib()
lib.smc_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
_lib.check(lib.smc_debug_trace(pf._f, buf.ctypes.data_as(ctypes.c_void_p)))
wide = two_level_wide = log2N <= 20 and scheme != "multinomial" and not os.environ.get("SMC_NO_WIDE")
two_level = not os.environ.get("SMC_FLAT_CDF")
for name, st, labels in (
        ("k_ancestors2" if two_level else "k_ancestors<true>", buf[nparts * 8:].reshape(ntiles, 8),
         ["start", "t + loads", "local cdf", "partials reduced", "tile shares", "counts", "end"] if two_level
         else ["start", "record", "q ready", "published", "prefix known", "counts", "end"]),
        ("k_propagate", buf[:nparts * 8].reshape(nparts, 8),
         ["start", "record", "loads+normals", "stores issued", "partial written"] if two_level
         else ["start", "record", "loads+normals", "stores issued", "wg reduced", "shard ticket",
               "top ticket", "finalised"])):
    raw = st
    st = st[st[:, 0] > 0]                                       # (k_ancestors2w: one row per workgroup of TPW tiles)
    if wide and name.startswith("k_ancestors2"):
        name, labels = "k_ancestors2w", ["start", "t known", "max exchanged", "shares known", "counts", "end"]
    t0 = st[:, 0].min()
    print("%s: %d workgroups, last start +%.2f us" % (name, st.shape[0], (st[:, 0].max() - t0) / 100.0))
    for k, lab in enumerate(labels):
        col = st[:, k].astype(np.int64) - int(t0)
        col = col[st[:, k] >= t0]
        if col.size == 0:
            continue
        print("  %-16s n=%5d  min %6.2f  median %6.2f  p90 %6.2f  p99 %6.2f  max %6.2f us (wg %d)"
              % (lab, col.size, col.min() / 100.0, np.median(col) / 100.0, np.percentile(col, 90) / 100.0,
                 np.percentile(col, 99) / 100.0, col.max() / 100.0, int(np.argmax(st[:, k]))))
    if os.environ.get("TRACE_WHO"):
        # who is late?  per stamp: workgroups later than median + 1 us, by XCD (index % 8) and by index range
        idx = np.nonzero(raw[:, 0] > 0)[0]
        for k, lab in enumerate(labels):
            col = (st[:, k].astype(np.int64) - int(t0)) / 100.0
            late = col > np.median(col) + 1.0
            if late.sum() == 0:
                continue
            print("    late at %-16s %4d wgs; by xcd %s; by eighth of the grid %s; start of the late %.2f (all %.2f)"
                  % (lab, late.sum(), np.bincount(idx[late] % 8, minlength=8).tolist(),
                     np.bincount(idx[late] * 8 // raw.shape[0], minlength=8).tolist(),
                     np.median((st[late, 0].astype(np.int64) - int(t0)) / 100.0),
                     np.median((st[:, 0].astype(np.int64) - int(t0)) / 100.0)))
    if st.shape[0] > 1024:
        # several rounds of workgroups: what a workgroup spends between two of its own stamps
        print("  per workgroup, between consecutive stamps (median / p90 us):")
        prev = 0
        for k in range(1, len(labels)):
            ok = (st[:, k] >= st[:, prev]) & (st[:, k] > 0)
            if not ok.any():
                continue
            dlt = (st[ok, k].astype(np.int64) - st[ok, prev].astype(np.int64)) / 100.0
            print("    %-18s -> %-18s %6.2f / %6.2f" % (labels[prev], labels[k], np.median(dlt), np.percentile(dlt, 90)))
            prev = k
    if name == "k_ancestors<true>":
        d = (st[:, 3].astype(np.int64) - st[:, 2].astype(np.int64)) / 100.0
        order = np.argsort(-d)[:8]
        print("  slowest q-ready -> published:", [(int(i), float(d[i]), float((st[i, 0] - t0) / 100.0)) for i in order])
        w = (st[:, 4].astype(np.int64) - st[:, 3].astype(np.int64)) / 100.0
        print("  published -> prefix known by tile index quartile:", [round(float(np.median(w[q * 256:(q + 1) * 256])), 2) for q in range(ntiles // 256)])

# ---- the step's floor: where a step's time goes BETWEEN the kernels.  All stamps are wall_clock64()
# ticks of one 100 MHz counter, so differences across the two launches of the step are meaningful.
if two_level:
    import time
    A = buf[nparts * 8:].reshape(ntiles, 8)
    A = A[A[:, 0] > 0]
    P = buf[:nparts * 8].reshape(nparts, 8)
    P = P[P[:, 0] > 0]
    last = lambda M: max(int(M[:, k].max()) for k in range(M.shape[1]))
    a0, a1, p0, p1 = int(A[:, 0].min()), last(A), int(P[:, 0].min()), last(P)
    K = 400
    pf.sync()
    t0 = time.perf_counter()
    pf.step_async(K)
    pf.sync()
    period = (time.perf_counter() - t0) / K * 1e6
    ra, gap_ap, rp = (a1 - a0) / 100.0, (p0 - a1) / 100.0, (p1 - p0) / 100.0
    print("floor breakdown at N = 2^%d (%s), us: step period %.2f (K = %d, this build: the stamps cost a store each) =" % (log2N, scheme, period, K))
    print("  resampling launch, first workgroup's first stamp -> last workgroup's last stamp   %6.2f" % ra)
    print("  its last stamp -> k_propagate's first workgroup running (end of kernel: write-back, dispatch of the dependent launch, first wave)   %6.2f" % gap_ap)
    print("  k_propagate, first stamp -> last stamp                                            %6.2f" % rp)
    print("  k_propagate's last stamp -> the NEXT step's first workgroup (period - the three above)   %6.2f" % (period - ra - gap_ap - rp))
