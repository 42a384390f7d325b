"""Round-4 paths in the combinations the test-suite does not spell out: hipGraph replay, history slots, islands,
the resident and the k_reduce2 grids -- each against the narrow / round-robin / literal variants, bit for bit."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import particles_amd as pa
from particles_amd import kalman, state_space_models as ssm
rng = np.random.RandomState(3)
T = 24
y = [np.array([v]) for v in np.cumsum(rng.standard_normal(T)) * 0.5]
SW = ("SMC_NO_WIDE", "SMC_WIDE4", "SMC_NO_XCD_CHUNKS", "SMC_STRICT_LITERAL")
def run(env, **kw):
    for k in SW: os.environ.pop(k, None)
    os.environ.update(env)
    pf = pa.SMC(fk=kw.pop("fk"), seed=9, collect="off", **kw)
    pf.run()
    return np.array(pf.A), np.array(pf.X), pf.logLts_islands.copy(), pf._summ().copy()
def same(a, b): return all(np.array_equal(p, q) for p, q in zip(a, b))
boot = ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y)
sv = ssm.Bootstrap(ssm=ssm.StochVol(), data=y)
apfb = ssm.AuxiliaryBootstrap(ssm=ssm.StochVol(), data=y)
n = 0
for fk, name in ((boot, "toy"), (sv, "sv")):
    for N in (1 << 15, 1 << 20, 1 << 21):
        for scheme in ("systematic", "stratified"):
            for extra in ({}, {"use_graph": True}, {"store_history": True}, {"n_islands": 3} if N < (1 << 20) else {"ESSrmin": 1.0}):
                base = run({}, fk=fk, N=N, resampling=scheme, **extra)
                for env in ({"SMC_NO_WIDE": "1"}, {"SMC_WIDE4": "1"}, {"SMC_NO_XCD_CHUNKS": "1"}, {"SMC_NO_WIDE": "1", "SMC_NO_XCD_CHUNKS": "1"}):
                    assert same(base, run(env, fk=fk, N=N, resampling=scheme, **extra)), (name, N, scheme, extra, env)
                    n += 1
print("two-level variants: %d comparisons equal" % n)
for N in (3000, 1 << 16, (1 << 18) + 5):
    for scheme in ("systematic", "stratified", "multinomial"):
        for M in (1, 3):
            a = run({}, fk=boot, N=N, resampling=scheme, strict_ancestors=True, n_islands=M, ESSrmin=0.9)
            b = run({"SMC_STRICT_LITERAL": "1"}, fk=boot, N=N, resampling=scheme, strict_ancestors=True, n_islands=M, ESSrmin=0.9)
            assert same(a, b), (N, scheme, M)
print("strict: emulation == literal walk, islands and schemes")
for N in (800, 1 << 14, 100000):
    a = run({}, fk=apfb, N=N, n_islands=2 if N > 1024 else 1)
    b = run({"SMC_NO_XCD_CHUNKS": "1"}, fk=apfb, N=N, n_islands=2 if N > 1024 else 1)
    assert same(a, b) and np.all(np.isfinite(a[2])), N
print("AuxiliaryBootstrap: fused at N = 800 / 2^14 / 10^5, islands")
