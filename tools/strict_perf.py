"""The reference's sequential fp64 CDF on the device: the parallel emulation (csrc/smc_seqsum.h) against the literal
one-lane walk -- equal bit for bit on every input, and what each costs; then the filter's strict mode
(SMC_FLAG_STRICT_ANCESTORS) on C2 with either.     python tools/strict_perf.py"""
import ctypes
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                        # noqa: E402
import particles_amd as pa                                          # noqa: E402
from particles_amd import _lib, kalman, state_space_models as ssm   # noqa: E402
from particles_amd._lib import DeviceArray, check, lib, ctx         # noqa: E402


def seq(d, S, N, mode, count=False):
    c = ctypes.c_int64(-1)
    check(lib().smc_seq_prefix_sums(ctx().h, d.ptr, N, S.ptr, mode, ctypes.byref(c) if count else None))
    return c.value


rng = np.random.default_rng(1)
for log2N in (16, 20, 22):
    N = 1 << log2N
    deg = np.exp(-60.0 + rng.standard_normal(N)); deg[rng.choice(N, 8, replace=False)] = 1.0
    cases = {"lognormal": np.exp(3 * rng.standard_normal(N)), "uniform": np.ones(N), "skewed": np.exp(40 * rng.standard_normal(N)),
             "degenerate": deg,
             "zeros": rng.random(N) * (rng.random(N) > 0.3), "collapsed": np.eye(1, N, N // 3)[0] + 0.0,
             "dyadic": rng.integers(0, 2 ** 30 // N + 2, size=N).astype(np.float64)}
    for name, w in cases.items():
        W = w / w.sum()
        d = DeviceArray.from_numpy(W)
        Sa, Sb = DeviceArray((N,)), DeviceArray((N,))
        fb = seq(d, Sa, N, 0, count=True)
        a = Sa.get()
        nseq = seq(d, Sa, N, 2, count=True)
        a2 = Sa.get()
        seq(d, Sb, N, 1)
        b = Sb.get()
        same = np.array_equal(a.view(np.uint64), b.view(np.uint64)) and np.array_equal(a2.view(np.uint64), b.view(np.uint64))
        t = {}
        for mode in (0, 2, 1):
            ts = []
            for _ in range(9 if mode != 1 else 3):             # (one call per device synchronisation; the median)
                ctx().sync()
                t0 = time.perf_counter()
                seq(d, Sa, N, mode)
                ctx().sync()
                ts.append(time.perf_counter() - t0)
            t[mode] = float(np.median(ts)) * 1e6
        print("N=2^%d %-10s %s  two launches (%s) %8.1f us | tile walk alone (%4d of %5d tiles exact) %8.1f us | literal walk %9.1f us"
              % (log2N, name, "EQUAL" if same else "DIFFERENT (%d)" % int((a != b).sum()),
                 "EXACT PATH     " if fb < 0 else "%3d exceptions" % fb, t[0], nseq, N // 1024, t[2], t[1]), flush=True)
        assert same

# ---- the filter's strict mode on C2 (ToySSM, N = 2^20, systematic) and C3 (StochVol, N = 2^22)
T = 220
y = bench.synthetic_data(T)
res = {}
for wl, mk, N, essr in (("C2", lambda: kalman.ToySSM(0.2), 1 << 20, 0.5), ("C3", lambda: ssm.StochVol(), 1 << 22, 1.0)):
    for scheme in (("systematic",) if wl == "C2" else ("systematic", "stratified", "multinomial")):
        fk = ssm.Bootstrap(ssm=mk(), data=y)
        for strict in (True, False):
            pf = pa.SMC(fk=fk, N=N, seed=5, collect="off", strict_ancestors=strict, resampling=scheme, ESSrmin=essr)
            pf.step_async(20)
            pf.sync()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                pf.step_async(40)
                pf.sync()
                ts.append((time.perf_counter() - t0) / 40)
            dt = float(np.median(ts))
            print("%s %-11s %-7s: %7.1f us per step, %6.2f G particle-steps/s" % (wl, scheme, "strict" if strict else "default", dt * 1e6, N / dt / 1e9),
                  flush=True)
if os.environ.get("SMC_STRICT_PERF_LITERAL"):
    fk = ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y[:30])
    for name, env in (("parallel", {}), ("literal", {"SMC_STRICT_LITERAL": "1"})):
        os.environ.pop("SMC_STRICT_LITERAL", None)
        os.environ.update(env)
        pf = pa.SMC(fk=fk, N=1 << 20, seed=5, collect="off", strict_ancestors=True)
        pf.run()
        res[name] = (np.array(pf.A), pf.logLt)
    assert np.array_equal(res["parallel"][0], res["literal"][0]) and res["parallel"][1] == res["literal"][1]
    print("strict ancestors: the two-launch CDF and the literal walk give the same run, bit for bit")
