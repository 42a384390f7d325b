"""CPU baseline of BASELINE.md section 3: the REFERENCE ITSELF (nchopin/particles imported
read-only from /root/reference) timed on the build container's host cores.

    PYTHONDONTWRITEBYTECODE=1 python tools/cpu_reference.py        # -> profiles/cpu_reference.json

/root/reference does not exist on the GPU box, so bench.py cannot run this there; it times the
oracle port live ("kind": "port") and prints this file's figures beside it, labelled as measured
in the build container.  Legs:
  * inputs: ToySSM(sigma=0.2) (README.md:66-78), np.random.seed(42); x, y = model.simulate(T);
    SMC(fk=Bootstrap(ssm, y), N, resampling='systematic', ESSrmin=0.5); run seed 123
  * timing: the reference's own pf.cpu_time (utils.py:81-89 around SMC.run, core.py:391),
    median of 3 repetitions; N = 2^20 runs T_timed steps (cost per step is flat in T)
  * 1 core, resampling.inverse_cdf (resampling.py:484-509) (a) as the pure-Python loop that
    `numba.jit` degrades to without numba, (b) bound to a gcc -O2 C restatement of the same loop
    (oracle/oracle.c orc_inverse_cdf_seq: element-identical output is asserted here) standing in
    for the numba-compiled function
  * all cores: particles.multiSMC(nruns=16, nprocs=<nproc>, collect='off', out_func=logLt) on
    N = 2^18, T = 20, after one warm-up call (core.py:431-518, utils.py:158-186), compiled
    inverse_cdf in the workers as well
"""
import ctypes
import json
import os
import platform
import statistics
import subprocess
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# numba is absent: a stand-in package whose jit() hands back the function unchanged, EXCEPT
# resampling.inverse_cdf, which it binds to the gcc -O2 restatement of the same loop
# (oracle.c orc_inverse_cdf_seq) -- what numba would have compiled.  It sits on PYTHONPATH so
# that multiSMC's worker processes get the same compiled inverse_cdf as the parent.
SHIM = os.path.join("/tmp", "smc_numba_c_shim")
os.makedirs(os.path.join(SHIM, "numba"), exist_ok=True)
_SHIM_SRC = ('''import ctypes, os
import numpy as np
_LIB = None
def _c_inverse_cdf(py):
    def inverse_cdf(su, W):
        global _LIB
        if _LIB is None:
            _LIB = ctypes.CDLL(%r)
            dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)
            _LIB.orc_inverse_cdf_seq.argtypes = [dp, dp, ctypes.c_int64, ctypes.c_int64, ip]
            _LIB.orc_inverse_cdf_seq.restype = ctypes.c_int64
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)
        su = np.ascontiguousarray(su, dtype=np.float64)
        W = np.ascontiguousarray(W, dtype=np.float64)
        A = np.empty(su.shape[0], dtype=np.int64)
        if _LIB.orc_inverse_cdf_seq(su.ctypes.data_as(dp), W.ctypes.data_as(dp), su.shape[0],
                                    W.shape[0], A.ctypes.data_as(ip)):
            raise IndexError("index out of bounds")
        return A
    inverse_cdf.py_func = py
    return inverse_cdf
def jit(*args, **kwargs):
    def wrap(f):
        return _c_inverse_cdf(f) if f.__name__ == "inverse_cdf" else f
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return wrap(args[0])
    return wrap
njit = jit
''' % os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
# (several worker processes of bench.py's all-cores leg import this module at once: the file is put in
#  place atomically, and only when it is not already the one wanted)
_shim_path = os.path.join(SHIM, "numba", "__init__.py")
if not (os.path.exists(_shim_path) and open(_shim_path).read() == _SHIM_SRC):
    _tmp = _shim_path + ".%d" % os.getpid()
    with open(_tmp, "w") as _fh:
        _fh.write(_SHIM_SRC)
    os.replace(_tmp, _shim_path)
subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
os.environ["PYTHONPATH"] = os.pathsep.join([SHIM, "/root/reference", ROOT, os.environ.get("PYTHONPATH", "")])
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.path[:0] = [SHIM, "/root/reference", ROOT]

import numpy as np  # noqa: E402
import particles  # noqa: E402
from particles import distributions as dists  # noqa: E402
from particles import resampling as rs  # noqa: E402
from particles import state_space_models as ssm  # noqa: E402


class ToySSM(ssm.StateSpaceModel):          # README.md:66-72
    default_params = {"sigma": 0.2}

    def PX0(self):
        return dists.Normal()

    def PX(self, t, xp):
        return dists.Normal(loc=xp)

    def PY(self, t, xp, x):
        return dists.Normal(loc=x, scale=self.sigma)


def c_inverse_cdf():
    """resampling.inverse_cdf as imported here: the compiled stand-in (its .py_func is the
    reference's own Python loop)."""
    return rs.inverse_cdf


def time_run(model, y, N, reps=3):
    out = []
    for r in range(reps):
        np.random.seed(123)
        pf = particles.SMC(fk=ssm.Bootstrap(ssm=model, data=y), N=N, resampling="systematic",
                           ESSrmin=0.5, collect="off")
        pf.run()
        out.append(pf.cpu_time)
    return statistics.median(out), out, pf.logLt


def logLt_of(pf):
    return pf.logLt


def main():
    res = {"host": {"cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"),
                    "nproc": os.cpu_count(), "python": platform.python_version(),
                    "numpy": np.__version__, "where": "build container (no GPU)"},
           "reference": "nchopin/particles v%s at /root/reference" % getattr(particles, "__version__", "?"),
           "legs": {}}
    model = ToySSM(sigma=0.2)
    c_icdf = c_inverse_cdf()
    py_icdf = c_icdf.py_func
    # the C stand-in computes what the reference's loop computes
    rng = np.random.default_rng(0)
    W = rng.random(5000); W /= W.sum()
    su = (rng.random() + np.arange(5000)) / 5000
    assert np.array_equal(py_icdf(su, W), c_icdf(su, W))

    for name, N, T in (("C1 N=1000 T=200", 1000, 200), ("C2-shape N=2^20 (30 steps timed)", 1 << 20, 30)):
        np.random.seed(42)
        x, y = model.simulate(T)
        for variant, fn in (("c_inverse_cdf", c_icdf), ("pure_python_inverse_cdf", py_icdf)):
            rs.inverse_cdf = fn
            med, allt, ll = time_run(model, y, N)
            res["legs"]["%s, 1 core, %s" % (name, variant)] = {
                "seconds_median": med, "seconds": allt, "N": N, "T": T, "cores": 1,
                "particle_steps_per_s": N * T / med, "ms_per_step": 1e3 * med / T, "logLt": float(ll)}
            print(name, variant, "%.3f s  %.2f M particle-steps/s" % (med, N * T / med / 1e6), flush=True)
    rs.inverse_cdf = c_icdf
    # all cores: multiSMC over its pool of worker processes (they import the same compiled
    # inverse_cdf through PYTHONPATH)
    nproc = os.cpu_count()
    N, T, nruns = 1 << 18, 20, 16
    np.random.seed(42)
    x, y = model.simulate(T)
    fk = ssm.Bootstrap(ssm=model, data=y)
    kw = dict(fk=fk, N=N, resampling="systematic", ESSrmin=0.5, nruns=nruns, nprocs=nproc,
              collect="off", out_func=logLt_of)
    particles.multiSMC(**dict(kw, nruns=nproc, N=1000))          # pool warm-up
    ts = []
    for r in range(3):
        t0 = time.perf_counter()
        out = particles.multiSMC(**kw)
        ts.append(time.perf_counter() - t0)
    med = statistics.median(ts)
    res["legs"]["C5-shape multiSMC 16 x (N=2^18, T=20), %d processes" % nproc] = {
        "seconds_median": med, "seconds": ts, "N": N, "T": T, "nruns": nruns, "cores": nproc,
        "particle_steps_per_s": nruns * N * T / med,
        "inverse_cdf": "compiled stand-in (C), parent and workers",
        "logLt_sd": float(np.std([o["output"] for o in out]))}
    print("multiSMC %d procs: %.2f s  %.1f M particle-steps/s" % (nproc, med, nruns * N * T / med / 1e6), flush=True)
    path = os.path.join(ROOT, "profiles", "cpu_reference.json")
    with open(path, "w") as fh:
        json.dump(res, fh, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
