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
    for k in ("SMC_EXACT_COUNTS", "SMC_TWO_LEVEL_MID", "SMC_FLAT_CDF", "SMC_NO_HEAVY"):
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
    for env in ({"SMC_EXACT_COUNTS": "1"}, {"SMC_TWO_LEVEL_MID": "1"}, {"SMC_NO_HEAVY": "1"}):
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
