"""Randomised cross-check of the step-loop paths on the GPU: for random sizes, island counts,
schemes, models and ESS thresholds the two-level path must be bit-identical with its fp64
band shortcut switched off (SMC_EXACT_COUNTS) and with k_reduce2 in front (SMC_TWO_LEVEL_MID),
with the heavy-parent list off (SMC_NO_HEAVY), and agree with the flat-Q62 path up to near-ties
(collapsing-weight models included).

    python tools/fuzz_paths.py [n_cases] [seed]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import particles_amd as pa                                           # noqa: E402
from particles_amd import kalman, state_space_models as ssm         # noqa: E402

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def run(env, mk, y, N, M, scheme, essr, seed):
    for k in ("SMC_EXACT_COUNTS", "SMC_TWO_LEVEL_MID", "SMC_FLAT_CDF", "SMC_NO_HEAVY", "SMC_SPLIT_REDUCE", "SMC_SPACING_3PASS", "SMC_NO_WIDE", "SMC_NO_XCD_CHUNKS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    pf = pa.SMC(fk=ssm.Bootstrap(ssm=mk(), data=y), N=N, n_islands=M, resampling=scheme, ESSrmin=essr,
                seed=seed)
    pf.run()
    return (np.array(pf.X), np.array(pf.A), np.array(pf.summaries.logLts), list(pf.summaries.rs_flags),
            np.array(pf.logLts_islands))


t0 = time.time()
flips = 0
for c in range(ncases):
    k = int(rng.integers(11, 21))
    N = 1 << k
    if rng.random() < 0.25:                 # not a power of two: the flat path only (heavy list on / off)
        N += int(rng.integers(1, N))
    M = int(rng.choice([1, 1, 2, 5])) if k <= 17 else 1
    scheme = str(rng.choice(["systematic", "stratified", "multinomial"]))     # (multinomial: counts by search)
    essr = float(rng.choice([0.3, 0.5, 0.9, 1.0]))
    T = int(rng.integers(5, 40))
    which = int(rng.integers(0, 6))
    mk = [lambda: kalman.ToySSM(0.2), lambda: kalman.ToySSM(0.01), lambda: ssm.StochVol(),
          lambda: kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=1.5),
          lambda: kalman.ToySSM(1e-5), lambda: kalman.ToySSM(1e-8)][which]       # 4, 5: collapsing weights
    y = [np.array([v]) for v in np.cumsum(rng.standard_normal(T)) * (0.3 if which == 2 else 1.0)]
    seed = int(rng.integers(1, 1 << 30))
    base = run({}, mk, y, N, M, scheme, essr, seed)
    envs = [{"SMC_EXACT_COUNTS": "1"}, {"SMC_TWO_LEVEL_MID": "1"}, {"SMC_NO_HEAVY": "1"}, {"SMC_NO_WIDE": "1"}, {"SMC_NO_XCD_CHUNKS": "1"}]
    if scheme == "multinomial":             # the island's reduction as a launch of its own; uniform_spacings in three passes
        envs += [{"SMC_SPLIT_REDUCE": "1"}, {"SMC_SPACING_3PASS": "1"}]
    for env in envs:
        oth = run(env, mk, y, N, M, scheme, essr, seed)
        assert np.array_equal(base[0], oth[0]) and np.array_equal(base[1], oth[1]), (c, env, N, M, scheme)
        assert np.array_equal(base[2], oth[2]) and base[3] == oth[3] and np.array_equal(base[4], oth[4]), (c, env)
    flat = run({"SMC_FLAT_CDF": "1"}, mk, y, N, M, scheme, essr, seed)
    same = np.array_equal(base[1], flat[1]) and np.array_equal(base[0], flat[0])
    if not same:
        flips += 1
        assert base[3] == flat[3] or abs(base[2][-1] - flat[2][-1]) < 0.5, (c, "flags differ", N, scheme)
    tol = 0.05 * np.sqrt(T) + 1e-9 if which < 4 else 1e-6 * abs(flat[2][-1]) + 1.0
    assert np.all(np.isfinite(base[2])) and abs(base[2][-1] - flat[2][-1]) < tol, \
        (c, N, M, scheme, base[2][-1], flat[2][-1])
    print("case %3d: N~2^%-2d M=%d %-10s ESSr=%.1f T=%-2d model %d  resampled %2d/%-2d  %s"
          % (c, k, M, scheme, essr, T, which, sum(base[3]), T, "== flat" if same else "near-tie vs flat"),
          flush=True)
print("ok: %d cases, %d with a near-tie difference to the flat path, %.1f s" % (ncases, flips, time.time() - t0))

# ---- SQMC: the fused loop against the operator path on the same points (random sizes, models, islands off)
from particles_amd import _lib, resampling as rs                      # noqa: E402
nsq = max(4, ncases // 8)
for k in ("SMC_EXACT_COUNTS", "SMC_TWO_LEVEL_MID", "SMC_FLAT_CDF", "SMC_NO_HEAVY", "SMC_SPLIT_REDUCE", "SMC_SPACING_3PASS", "SMC_NO_WIDE", "SMC_NO_XCD_CHUNKS"):
    os.environ.pop(k, None)
rs.set_rng("philox")
ties = 0
try:
    for c in range(nsq):
        which = int(rng.integers(0, 5))
        d = [1, 1, 1, 2, 3][which]
        k = int(rng.integers(5, 19)) if d == 1 else int(rng.integers(6, 17))      # (d = 1: flat step below 2^11)
        N, T = 1 << k, int(rng.integers(3, 9))
        mk, cls = [(lambda: kalman.ToySSM(0.2), ssm.Bootstrap), (lambda: ssm.StochVol(), ssm.Bootstrap),
                   (lambda: kalman.LinearGauss(rho=0.9, sigmaX=1.0, sigmaY=0.3), ssm.GuidedPF),
                   (lambda: kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=2), ssm.Bootstrap),
                   (lambda: kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=3), ssm.GuidedPF)][which]
        y = [np.atleast_1d(v) for v in (rng.standard_normal((T, d)) * 0.5)]
        seed = int(rng.integers(1, 1 << 30))
        out = []
        for fused in (True, False):
            _lib.FUSED_SQMC[0] = fused
            pa.seed(seed)
            pf = pa.SMC(fk=cls(ssm=mk(), data=y), N=N, qmc=True, collect="off")
            assert pf._fused == fused, (c, N, d)
            pf.run()
            out.append((np.asarray(pf.X), np.asarray(pf.A), float(pf.logLt)))
        _lib.FUSED_SQMC[0] = True
        same = np.array_equal(out[0][1], out[1][1])
        if same:
            assert np.allclose(out[0][0], out[1][0], rtol=1e-11, atol=1e-11), (c, N, d, "X")
        else:                                   # (a tie between the two exact CDFs / a last bit of a weight)
            ties += 1
        assert abs(out[0][2] - out[1][2]) < (1e-10 * max(1.0, abs(out[0][2])) if same else 0.05), (c, N, d, out[0][2], out[1][2])
        print("sqmc %2d: N=2^%-2d d=%d model %d T=%d  %s" % (c, k, d, which, T, "== operators" if same else "near-tie"), flush=True)
finally:
    _lib.FUSED_SQMC[0] = True
    rs.set_rng("numpy")
print("ok: %d SQMC cases, %d with a near-tie difference to the operator path" % (nsq, ties))
