#!/usr/bin/env python
"""Headline benchmark: particle-steps/s of the bootstrap filter of BASELINE.json
config C2 -- ToySSM (d=1 linear Gaussian), N = 2^20 particles, systematic
resampling, ESSrmin = 0.5 -- on N GPUs of one node.

    python bench.py --gpus 1 --steps 1000 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...

A "step" is one time step of the filter (resample decision, resampling,
propagation, weighting, evidence increment) over all N particles of the rank's
filter.  One process per GPU; each rank runs an independent filter on the same
data with its own Philox island id (weak scaling: total work = n_gpus * N * K).
The path has NO data-path collective, so the timed region contains none: the one
collective of the path -- the all-gather of the per-island log-evidences over RCCL
(smc_comm_*), once per RUN of T steps -- is timed on its own and reported beside the
value (`evidence_gather_ms`).  Inputs are resident in HBM before the timed region starts.

The timed region of exactly K steps (barrier + device sync on both sides) is repeated R times
on consecutive stretches of the same run (R such that about 10^4 steps are timed in all);
`ms_per_step` / `value` are the MEDIAN repetition, the spread is reported beside them -- a
K = 20 region lasts half a millisecond and a single shot of it is noise.

Rank 0 prints ONE JSON line (see README / task contract), including
  roofline        -- the longer of the step's two kernels (resampling: k_ancestors2 /
                     k_ancestors; propagate: k_propagate): algorithmic bytes per launch over
                     its average duration, BOTH measured with HIP events on the filter's stream
                     (smc_filter_kernel_ms), against the 8 TB/s HBM peak of MI355X;
  cpu_baseline    -- N=1, rank 0: the CPU path on THIS host's cores (nproc stated): one core and
                     all cores (independent runs over worker processes, the shape of
                     multiSMC(nruns, nprocs=nproc), core.py:431 / utils.py:158-186).  kind =
                     "reference" when nchopin/particles is importable from /root/reference (build
                     container: its own pf.cpu_time), else "port": the NumPy restatement of its path
                     (oracle/), with the reference's figures from profiles/cpu_reference.json beside it;
  other_workloads -- N=1, default workload only: one bounded measurement of each of the other
                     BASELINE.json configs (C3 x three schemes, C4, C4 collapsed weight, C5) with
                     value, ms_per_step, step_frac and the dominant kernel's roofline fraction.

A single BASELINE.json config as the headline of the line: --workload c3 (StochVol N=2^22; --scheme),
c4 (d=32 guided), c5 (32 islands x 2^18 per GPU; `--workload c5 --gpus 8` is the 256-island run).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_PEAK_TF = 78.6              # dense fp64 matrix (= vector) peak
BYTES_PREPARE = 16.0             # k_ancestors: read lw, write A  (beyond 2048 workgroups per launch
                                 # k_prepare reads lw once more for the tile totals: + 8 B)
REFERENCE_DIR = "/root/reference"


def synthetic_data(T, sigma=0.2, seed=42):
    """BASELINE.md section 3's inputs: ToySSM(sigma) (README.md:66-78), np.random.seed(42);
    x, y = model.simulate(T) -- our simulate() consumes numpy's legacy stream exactly as the
    reference's does (state_space_models.py:283-323), so these are the reference's data."""
    from particles_amd import kalman
    np.random.seed(seed)
    x, y = kalman.ToySSM(sigma).simulate(T)
    return y


def kalman_loglik_toy(y, sigma=0.2):
    """Exact log-likelihood of ToySSM(sigma) (README.md:66-72: X_0 ~ N(0,1), X_t = X_{t-1} + N(0,1), Y_t = X_t +
    N(0, sigma^2)) by the scalar Kalman recursions (kalman.py:169-229 specialised to d = 1): what every particle
    estimate of log p(y_{0:n}) in the line -- the GPU's, the CPU baseline's -- estimates."""
    m, P, ll = 0.0, 1.0, 0.0
    for t, yt in enumerate(np.asarray(y, dtype=np.float64).ravel()):
        if t:
            P = P + 1.0
        S = P + sigma * sigma
        ll += -0.5 * (np.log(2.0 * np.pi * S) + (yt - m) ** 2 / S)
        Kg = P / S
        m, P = m + Kg * (yt - m), (1.0 - Kg) * P
    return float(ll)


def leg_key(wl):
    """Name of a workload in `other_workloads` and in profiles/traffic_<key>.json."""
    if wl.get("qmc"):
        return "sqmc"
    sfx = "_strict" if wl.get("strict") else ""
    if wl["name"] == "c3":
        return "c3_" + wl["scheme"] + sfx
    if wl["name"] == "c4":
        return "c4_collapsed" if wl["collapsed"] else ("c4_dense" if wl.get("dense") else "c4")
    return wl["name"] + sfx


def committed_profile(wl, kernel, profiles=None):
    """What the committed rocprofv3 passes of this workload's own command line say about `kernel` (`a+b`: the sum over
    the launches named): HBM bytes per launch from the PMC passes (tools/gpu_profile_all.sh -> tools/summarise_prof.py:
    FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as the MI355X guide prescribes for gfx950) and the
    average duration in the --kernel-trace --stats pass.  PMC counters cannot be collected from inside the process, so the
    figures are reported only when the committed record's `config` is the configuration being run; otherwise None.
    Returns {"bytes", "avg_us" (None in records older than round 5), "source"} or None."""
    path = os.path.join(profiles or os.path.join(ROOT, "profiles"), "traffic_%s.json" % leg_key(wl))
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        rec = json.load(fh)
    if rec.get("source_hash") and rec["source_hash"] != _source_hash():
        return None          # (a record of kernels whose source has changed since: not this tree's)
    cfg = rec.get("config", {})
    if wl["Nlabel"] != "2^%d" % wl["log2N"] or \
            (cfg.get("log2N"), cfg.get("islands"), cfg.get("scheme")) != (wl["log2N"], wl["islands"], wl["scheme"]) or \
            bool(cfg.get("collapsed")) != bool(wl["collapsed"]) or bool(cfg.get("qmc")) != bool(wl.get("qmc")) or \
            bool(cfg.get("strict")) != bool(wl.get("strict")) or bool(cfg.get("dense")) != bool(wl.get("dense")):
        return None
    total, found, us, us_ok = 0.0, 0, 0.0, True
    base = lambda name: name.replace("void ", "").split("<")[0].split("(")[0].strip()
    steps = max([d.get("launches") or 0 for name, d in rec["kernels"].items() if base(name).startswith("k_propagate")] + [0])
    for part in kernel.split("+"):
        part = part.split("<")[0].split("(")[0].split(" [")[0].strip()
        if part == "k_rs_sort":         # the radix sort is several launches per step (k_rs_hist / _scan / _scatter)
            parts = [d for name, d in rec["kernels"].items() if base(name).startswith("k_rs_")]
            if parts and steps and all(d.get("launches") for d in parts):
                total += sum(d["hbm_bytes_per_launch"] * d["launches"] for d in parts) / steps
                if all(d.get("avg_us") for d in parts):
                    us += sum(d["avg_us"] * d["launches"] for d in parts) / steps
                else:
                    us_ok = False
                found += 1
            continue
        hits = [d for name, d in rec["kernels"].items() if part == base(name)]
        if hits:
            d = max(hits, key=lambda h: h["hbm_bytes_per_launch"])   # (instantiations of one template: the one that carries the step)
            total += d["hbm_bytes_per_launch"]
            if d.get("avg_us"):
                us += d["avg_us"]
            else:
                us_ok = False
            found += 1
    if not found:
        return None
    return {"bytes": total, "avg_us": us if us_ok and us > 0 else None,
            "source": "rocprofv3 PMC, profiles/%s (tree %s, source %s)" % (rec.get("summary", ""), rec.get("tree", "?"),
                                                                          rec.get("source_hash", "unstamped"))}


_SRC_HASH = []


def _source_hash():
    if not _SRC_HASH:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from source_hash import source_hash
        _SRC_HASH.append(source_hash())
    return _SRC_HASH[0]


def measured_traffic(wl, kernel, profiles=None):
    """(HBM bytes per launch, source) of `kernel` from the committed PMC passes, or None (see committed_profile)."""
    cp = committed_profile(wl, kernel, profiles)
    return (cp["bytes"], cp["source"]) if cp else None


def add_profile_fractions(rf, wl, profiles=None):
    """roofline.frac is measured live with HIP events; the two fractions a reader can RECOMPUTE from profiles/ go beside
    it: frac_rocprof = the same algorithmic bytes (or flop) over the kernel's average duration in the committed
    rocprofv3 --kernel-trace --stats pass, frac_physical = the HBM bytes the committed PMC passes counted over that
    duration (hbm-bound legs).  Both null until a round-5 record (with avg_us) of this workload is committed."""
    rf.update({"frac_rocprof": None, "frac_physical": None, "rocprof_kernel_us": None})
    cp = committed_profile(wl, rf["kernel"], profiles)
    if not cp:
        return rf
    rf["traffic"], rf["traffic_source"] = cp["bytes"], cp["source"]
    if cp["avg_us"]:
        sec = cp["avg_us"] * 1e-6
        rf["rocprof_kernel_us"] = cp["avg_us"]
        if rf["bound"] == "mfma":
            rf["frac_rocprof"] = rf["launch_flop"] / sec / 1e12 / FP64_PEAK_TF
        else:
            rf["frac_rocprof"] = rf["launch_bytes"] / sec / 1e9 / HBM_PEAK_GBS
            rf["frac_physical"] = cp["bytes"] / sec / 1e9 / HBM_PEAK_GBS
        rf["fractions_note"] = ("frac: HIP events, this run; frac_rocprof: launch_bytes (launch_flop) / rocprof_kernel_us / peak; "
                                "frac_physical: traffic / rocprof_kernel_us / peak -- rocprof_kernel_us and traffic are the "
                                "committed record's (%s)" % cp["source"])
    return rf


# ----------------------------------------------------------------------------------------------
# CPU baseline
# ----------------------------------------------------------------------------------------------
def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_DIR, "particles"))


def _cpu_worker(kind, N, nsteps, nruns, gate, Tdata=0, index=0):
    """One worker PROCESS of the CPU baseline (bench.py --cpu-worker ...): `nruns` independent
    bootstrap filters of N particles over the first `nsteps` observations, one after the other,
    on one core; run r of worker `index` is seeded 123 + index * nruns + r -- distinct seeds over the whole
    pool, as multiSMC's workers get them (utils.py:189-213).  gate: a directory -- the worker drops a `ready.<pid>` file when its imports are
    done and starts when the parent creates `go` (so that all workers run at the same time).
    Prints {"t0", "t1", "seconds", "logLt"}; the parent aggregates."""
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
    if kind == "reference":
        sys.dont_write_bytecode = True
        import tools.cpu_reference as cr            # numba shim + compiled inverse_cdf + /root/reference
        model = cr.ToySSM(sigma=0.2)
        np.random.seed(42)
        x, y = model.simulate(max(nsteps, Tdata))     # (the GPU run's series: simulate(T) draws depend on T)
        y = y[:nsteps]

        def one(seed):
            np.random.seed(seed)
            pf = cr.particles.SMC(fk=cr.ssm.Bootstrap(ssm=model, data=y), N=N, resampling="systematic",
                                  ESSrmin=0.5, collect="off")
            pf.run()
            return pf.cpu_time, float(pf.logLt)       # the reference's own timer (utils.py:81-89)
    else:
        from oracle import smc_oracle as orc
        y = synthetic_data(max(nsteps, Tdata))        # (the GPU run's series, of which the first nsteps are filtered)

        def one(seed):
            np.random.seed(seed)
            t0 = time.perf_counter()
            out = orc.run_filter(orc.ToySSM(0.2), y[:nsteps], N, "systematic", 0.5)
            return time.perf_counter() - t0, float(out["final_logLt"])
    if gate:
        open(os.path.join(gate, "ready.%d" % os.getpid()), "w").close()
        t_wait = time.time()
        while not os.path.exists(os.path.join(gate, "go")) and time.time() - t_wait < 300:
            time.sleep(0.002)
    t0 = time.time()
    secs, lls = 0.0, []
    for r in range(nruns):
        s, ll = one(123 + index * nruns + r)
        secs += s
        lls.append(ll)
    print(json.dumps({"t0": t0, "t1": time.time(), "seconds": secs, "logLt": lls}), flush=True)


def _spawn_workers(kind, N, nsteps, runs_per_worker, nworkers, Tdata=0):
    import tempfile
    gate = tempfile.mkdtemp(prefix="smc_cpu_gate_") if nworkers > 1 else ""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-worker", kind, str(N), str(nsteps),
           str(runs_per_worker), gate, str(int(Tdata))]
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    procs = [subprocess.Popen(cmd + [str(i)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env)
             for i in range(nworkers)]
    if gate:
        t_wait = time.time()        # all imports done (or a worker died / 120 s passed): go
        while time.time() - t_wait < 120:
            if sum(f.startswith("ready.") for f in os.listdir(gate)) >= nworkers or any(p.poll() is not None for p in procs):
                break
            time.sleep(0.01)
        open(os.path.join(gate, "go"), "w").close()
    outs, failed = [], []
    deadline = time.time() + 240.0
    for p in procs:
        try:
            so, se = p.communicate(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            so, se = p.communicate()
            failed.append("timeout")
            continue
        if p.returncode != 0:                       # (a negative code: killed by a signal, e.g. out of memory)
            failed.append("rc=%d %s" % (p.returncode, se[-200:].strip()))
            continue
        outs.append(json.loads(so.strip().splitlines()[-1]))
    if gate:
        import shutil
        shutil.rmtree(gate, ignore_errors=True)
    if not outs:
        raise RuntimeError("every cpu worker failed: " + "; ".join(failed[:3]))
    if failed:
        print("bench.py: %d of %d cpu workers failed (%s): the leg counts the others" % (len(failed), nworkers, failed[0]),
              file=sys.stderr)
    return outs


def cpu_baseline(N, nsteps, all_cores=True, kind=None, Tdata=0):
    """CPU legs beside the GPU number, on this host's cores.
    kind "reference": nchopin/particles itself (importable only where /root/reference exists: the
    build container), timed by its own pf.cpu_time, `inverse_cdf` bound to its gcc -O2 restatement
    (numba is absent); kind "port": oracle.run_filter, the NumPy restatement of that path."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
    if kind is None:
        kind = "reference" if reference_available() else "port"
    nproc = host_cores()
    ref_file = os.path.join(ROOT, "profiles", "cpu_reference.json")
    committed = None
    if os.path.exists(ref_file):
        with open(ref_file) as fh:
            rec = json.load(fh)
        committed = {"host": rec["host"], "source": "profiles/cpu_reference.json (tools/cpu_reference.py)",
                     "legs": {k: {"particle_steps_per_s": v["particle_steps_per_s"], "cores": v["cores"]}
                              for k, v in rec["legs"].items()}}
    what = ("particles.SMC(...).run(), pf.cpu_time; inverse_cdf bound to its gcc -O2 restatement (numba absent)"
            if kind == "reference" else
            "oracle.run_filter (NumPy restatement of particles.SMC + C inverse_cdf)")
    sample = ("%s; N=2^%d, first %d steps of the same data (np.random.seed(42); simulate), run seed 123; "
              "cost per step is flat in T" % (what, int(np.log2(N)), nsteps))
    one = _spawn_workers(kind, N, nsteps, 1, 1, Tdata)[0]
    out = {"value": N * nsteps / one["seconds"], "unit": "particle-steps/s", "cores": 1, "kind": kind,
           "sample": sample, "seconds": one["seconds"], "logLt": one["logLt"][0],
           "host": {"nproc": nproc, "cpu": _cpu_name()},
           "reference_build_container": committed}
    if kind == "port":
        # (so that nobody reads the port's figure as the reference's: the restatement has none of scipy.stats'
        #  per-call overhead and is several times FASTER than nchopin/particles itself on the same core)
        out["note"] = ("kind 'port' = the oracle's NumPy restatement of particles.SMC (the reference tree does not exist on "
                       "this box); it is faster than the reference itself -- the reference's own figures, timed in the "
                       "build container, are under reference_build_container")
    if all_cores and nproc > 1:
        # independent runs over worker processes: what multiSMC(nruns, nprocs=nproc) does (core.py:431,
        # utils.py:158-186): runs of N = 2^18 (BASELINE.md section 3's shape), at least 16 in all and at
        # least one per core, long enough (about 10^8 particle-steps per worker) that process start-up
        # and scheduling noise do not matter; the workers start together once all have imported
        # (the NumPy path streams five arrays per step through DRAM: beyond a few dozen processes a
        #  host's memory bandwidth, not its core count, is the limit -- and every worker is a Python
        #  interpreter with numpy and scipy loaded: at most SMC_BENCH_CPU_WORKERS = 64 of them)
        Na = min(N, 1 << 18)
        Ta = max(1, min(nsteps, int(1e8 // Na) if kind == "port" else int(3e7 // Na)))
        nw = max(1, min(nproc, int(os.environ.get("SMC_BENCH_CPU_WORKERS", "64"))))
        per = max(1, -(-16 // nw))
        try:
            ws = _spawn_workers(kind, Na, Ta, per, nw)
            wall = max(w["t1"] for w in ws) - min(w["t0"] for w in ws)
            out["all_cores"] = {
                "value": len(ws) * per * Na * Ta / wall, "unit": "particle-steps/s", "cores": len(ws),
                "host_nproc": nproc, "kind": kind, "seconds": wall, "runs": len(ws) * per,
                "sum_of_worker_rates": float(sum(per * Na * Ta / (w["t1"] - w["t0"]) for w in ws)),
                "start_skew_s": max(w["t0"] for w in ws) - min(w["t0"] for w in ws),
                "sample": "%d independent runs (seeds 123 ... %d, one per run) of (N=2^%d, %d steps) over %d worker "
                          "processes, one per core, on a host of %d (multiSMC(nruns=%d, nprocs=%d) shape); value = all "
                          "work / (last end - first start)"
                          % (len(ws) * per, 122 + nw * per, int(np.log2(Na)), Ta, len(ws), nproc, len(ws) * per, len(ws)),
                "logLt_sd": float(np.std([l for w in ws for l in w["logLt"]])),
                "distinct_logLt": len(set(l for w in ws for l in w["logLt"]))}
        except Exception as e:          # the leg is a reported baseline: it must not cost the line
            out["all_cores"] = {"error": "%s: %s" % (type(e).__name__, e), "cores": nw, "host_nproc": nproc}
    return out


def _cpu_name():
    try:
        return open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")
    except Exception:
        return "unknown"


# ----------------------------------------------------------------------------------------------
# workloads (BASELINE.json configs)
# ----------------------------------------------------------------------------------------------
def make_workload(name, T, scheme="systematic", log2N=None, N=0, islands=1, essrmin=None, collapsed=False, qmc=False,
                  strict=False, dense=False):
    from particles_amd import kalman
    from particles_amd import state_space_models as ssm
    d = 1
    if name == "c3":          # StochVol, N = 2^22, ESSrmin = 1: every step resamples
        log2N = 22 if log2N is None else log2N
        essrmin = 1.0 if essrmin is None else essrmin
        rng = np.random.RandomState(42)
        model = ssm.StochVol()
        x = np.empty(T)
        x[0] = model.mu + model.sig0() * rng.standard_normal()
        for t in range(1, T):
            x[t] = model.EXt(x[t - 1]) + model.sigma * rng.standard_normal()
        y = [np.array([v]) for v in np.exp(0.5 * x) * rng.standard_normal(T)]
        fk = ssm.Bootstrap(ssm=model, data=y)
        label = "C3: StochVol d=1 bootstrap filter"
    elif name == "c4":        # MVLinearGauss_Guarniero d = 32, guided, N = 2^20
        d = 32
        log2N = 20 if log2N is None else log2N
        rng = np.random.RandomState(42)
        model = kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=d)
        x = np.zeros(d)
        y = []
        for t in range(T):
            x = (model.F @ x if t else np.zeros(d)) + rng.standard_normal(d)
            y.append((x + rng.standard_normal(d)).reshape(1, d))
        fk = ssm.GuidedPF(ssm=model, data=y)
        label = "C4: MVLinearGauss_Guarniero d=32 guided filter" + (" (collapsed proposal weight)" if collapsed else "") + \
                (" [SMC_PATH_MV_DENSE: the model's diagonal factors applied as dense MFMA products all the same]" if dense else "")
    else:
        if name == "c5":      # one GPU's share of 256 islands x 2^18
            log2N = 18 if log2N is None else log2N            # (smaller: functional tests on the emulator)
            islands = 32 if islands == 1 else islands
        log2N = 20 if log2N is None else log2N
        fk = ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=synthetic_data(T))
        label = "C2: ToySSM d=1 linear-Gaussian bootstrap filter" if name == "c2" else \
                "C5: ToySSM d=1 bootstrap filter islands"
        if qmc:               # SMC(qmc=True) on C2's model and size (core.py:339-349), the fused SQMC step
            label = "SQMC (Sobol' points, sort, inverse CDF, ppf moves) on C2's model"
    return {"name": name, "fk": fk, "N": N if N > 0 else 1 << log2N, "log2N": log2N, "Nlabel": str(N) if N > 0 else "2^%d" % log2N,
            "islands": islands, "scheme": scheme, "essrmin": 0.5 if essrmin is None else essrmin, "d": d,
            "label": label + (" [strict_ancestors: the reference's sequential fp64 inverse_cdf, bit for bit]" if strict else ""),
            "collapsed": collapsed, "guided": name == "c4", "qmc": qmc, "strict": strict, "dense": bool(dense) and name == "c4"}


def make_filter(wl, rank=0, graph=False, profile=False):
    import particles_amd as pa
    from particles_amd import _lib
    from particles_amd import resampling as rs
    mode = _lib.RNG_MODE[0]
    if wl.get("qmc"):
        rs.set_rng("philox")              # (device-generated points: the fused SQMC step)
    if wl.get("dense"):                   # (a verification switch, read when the filter is created: _lib.path_flags)
        os.environ["SMC_MV_DENSE"] = "1"
    try:
        pf = pa.SMC(fk=wl["fk"], N=wl["N"], resampling=wl["scheme"], ESSrmin=wl["essrmin"], collect="off", seed=123,
                    n_islands=wl["islands"], island_offset=rank * wl["islands"], qmc=bool(wl.get("qmc")),
                    use_graph=graph and not profile, collapsed_proposal=wl["collapsed"],
                    strict_ancestors=bool(wl.get("strict")))
    finally:
        rs.set_rng(mode)
        os.environ.pop("SMC_MV_DENSE", None)
    if wl.get("qmc") and not pf._fused:
        raise RuntimeError("SQMC did not take the fused step (N = 2^k >= 2048 is required)")
    if profile:
        _lib.check(_lib.lib().smc_filter_profile(pf._f, 1))
    return pf


def time_steps(pf, K, W, R, grp=None):
    """W untimed steps, then R repetitions of: [barrier,] device sync, clock, K steps, device sync, clock
    [, barrier].  Returns the R local times."""
    pf.step_async(W)
    pf.sync()
    if grp:       # warm-up of the path's one collective too (RCCL sets its channels up lazily)
        grp.gather_evidence(pf.logLts_islands)
    dts = np.zeros(R)
    for r in range(R):
        if grp:
            grp.barrier()
        pf.sync()
        t0 = time.perf_counter()
        pf.step_async(K)
        pf.sync()
        dts[r] = time.perf_counter() - t0
        if grp:
            grp.barrier()
    return dts


def kernel_profile(wl, K, W, rank=0):
    """The same workload re-run with HIP events around every launch on the filter's stream (kept out of
    the timed region): average duration of the propagate kernel and of the resampling kernel(s)."""
    import ctypes
    from particles_amd import _lib
    pf = make_filter(wl, rank, profile=True)
    # (at least 60 untimed steps and 240 sampled ones whatever K is -- the data permitting: at the driver's K = 20 the
    #  20 sampled steps of round 3 gave seven samples per kind on a GPU that had just idled, and the dominant kernel
    #  read 4 % slower than in rocprofv3's trace of the same command)
    T = wl["fk"].T
    W = max(W, min(60, T // 4))
    pf.step_async(W)
    pf.sync()
    mv, pr, ns = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    _lib.check(_lib.lib().smc_filter_kernel_ms(pf._f, ctypes.byref(mv), ctypes.byref(pr),
                                               ctypes.byref(ns)))    # drop warm-up samples
    pf.step_async(max(1, min(max(K, 240), 4000, T - W)))
    _lib.check(_lib.lib().smc_filter_kernel_ms(pf._f, ctypes.byref(mv), ctypes.byref(pr), ctypes.byref(ns)))
    desc = ctypes.create_string_buffer(256)
    _lib.check(_lib.lib().smc_filter_describe(pf._f, desc, 256))
    del pf
    return mv.value, pr.value, ns.value, desc.value.decode()


def roofline(wl, step_GBs, mv_ms, rs_ms, nsamples, kernels):
    """The roofline object of one workload from the measured durations of its step's two parts."""
    N, isl, d = wl["N"], wl["islands"], wl["d"]
    names = kernels.split(" [")[0].split("+")
    rs_name = "+".join(k for k in names if not k.startswith("k_propagate"))
    mv_name = [k for k in names if k.startswith("k_propagate")][0]
    # two-level path: k_propagate also writes the tile CDF (8 B), k_ancestors2 reads it
    # instead of the log-weights: 16 d + 24 and 16 B; flat path: 16 d + 16 and 16 (+ 8) B;
    # either way SURVEY 8d's 16 d + 40 B per particle-step in all
    strict2 = "k_strict_classify" in kernels
    two = "k_ancestors2" in kernels or strict2
    # strict two-level step: k_strict_classify reads lw and writes the 8-byte integer prefix of every weight's rounding,
    # k_strict_search reads it and writes A: 32 B per particle (the sequential CDF itself is never written)
    # (k_strict_step, both in one launch: that prefix stays in registers -- 16 B, the tile-scale weights in, A out)
    rs_bytes = ((16.0 if "k_strict_step" in kernels else 32.0) if strict2
                else BYTES_PREPARE + (8.0 if "k_prepare" in kernels else 0.0)) * N * isl
    mv_bytes = (16.0 * d + 16.0 + (8.0 if two else 0.0)) * N * isl
    mv_ms = mv_ms if mv_ms > 0 else 1e-9          # (the emulator's events read 0)
    per = {mv_name: {"ms": mv_ms, "launch_bytes": mv_bytes, "achieved": mv_bytes / (mv_ms * 1e-3) / 1e9},
           rs_name: {"ms": rs_ms, "launch_bytes": rs_bytes,
                     "achieved": rs_bytes / (rs_ms * 1e-3) / 1e9 if rs_ms > 0 else None}}
    dom = mv_name if mv_ms >= rs_ms else rs_name
    ach = per[dom]["achieved"]
    out = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
           "traffic": None, "traffic_source": None, "kernel": dom, "kernel_ms": per[dom]["ms"],
           "launch_bytes": per[dom]["launch_bytes"], "samples": nsamples, "step_kernels": kernels,
           "per_kernel": per, "step_frac": step_GBs / HBM_PEAK_GBS}
    if wl["guided"] and d == 32:
        # GEMM-shaped kernel: priced against the dense fp64 matrix peak (MI355X spec 78.6 TFLOP/s, = its
        # fp64 vector peak).  72 MFMAs (v_mfma_f64_16x16x4: 2048 flop) per 16 particles
        # for the guided d=32 step, 44 with the collapsed weight.
        # "[diagonal factors]" (BASELINE C4's model: G = covX = covY = I): the three triangular factors are applied
        # element by element and 32 MFMAs per 16 particles remain (F xp, B xp) -- 4 096 flop against 528 B per
        # particle is BELOW the ridge (78.6 TF / 8 TB/s = 9.8 flop/B), so the bounding roofline of that form is HBM
        # ("report both bounds for C4"); what limits it in fact is VALU issue (mfma_frac, limiter)
        diag = "[diagonal factors]" in kernels
        nm = mfma_per_16(wl["collapsed"], diag)
        flop = nm * 2048.0 / 16.0 * N * isl
        tf = flop / (mv_ms * 1e-3) / 1e12
        if diag:
            out.update({"kernel": "k_propagate_mv", "kernel_ms": mv_ms, "launch_bytes": mv_bytes, "launch_flop": flop,
                        "achieved": per[mv_name]["achieved"], "frac": per[mv_name]["achieved"] / HBM_PEAK_GBS,
                        "mfma_achieved_TF": tf, "mfma_frac": tf / FP64_PEAK_TF, "mfma_per_16_particles": nm,
                        "limiter": "valu", "limiter_note": "16 Philox4x32-10 calls + Box-Muller pairs per particle (about 114 vector "
                        "instructions each) -- and an fp64 MFMA holds the SIMD's vector issue for its 64 cycles "
                        "(profiles/r15_mfma_shadow.txt), so matrix and vector time add"})
        else:
            out.update({"bound": "mfma", "achieved": tf, "peak": FP64_PEAK_TF, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TF,
                        "kernel": "k_propagate_mv", "kernel_ms": mv_ms, "launch_bytes": mv_bytes, "launch_flop": flop,
                        "hbm_achieved_GBs": per[mv_name]["achieved"], "mfma_per_16_particles": nm})
    return out


def mfma_per_16(collapsed, diag):
    """v_mfma_f64_16x16x4_f64 per 16 particles of k_propagate_mv's guided d = 32 step (smc_filter_mv.h): F xp and B xp
    (16 each), L_P z and L_X^-1 (x - m) (12 each, triangular), -(L_Y^-1 G) x (16); collapsed weight: B xp, -(L_S^-1 G F) xp,
    L_P z; with diagonal factors the triangular ones and -(L_Y^-1 G) are element-wise."""
    if diag:
        return 32
    return 44 if collapsed else 72


ROOFLINE_NOTE = (
    "algorithmic bytes in SURVEY 8d's accounting (int64 ancestors; they are stored as 32-bit words, so the "
    "kernels physically move 4 B less per particle each). Per particle-step, two-level path: k_propagate "
    "16 d + 24 B (read A, gather X; write X, lw and the tile's integer CDF), k_ancestors2 16 B (read that CDF, "
    "write A); flat path: k_propagate 16 d + 16 B, k_ancestors 16 B (read lw, write A; + 8 B for k_prepare's "
    "pass over lw beyond 2048 workgroups per launch): 16 d + 40 B in all. Both parts are measured with HIP "
    "events on the filter's stream in a separate pass over the same workload: steps are sampled in three "
    "kinds (whole step / up to the propagate launch / from there on), propagate = whole - first part, "
    "resampling = whole - second part, so the fixed ~4 us of an event interval cancels. `kernel` is the one "
    "that takes longer; step_frac = (16 d + 40) B x N / ms_per_step over the HBM peak (the whole step)")


def other_workloads(K=20, W=100, R=15, shrink=0):
    """One bounded measurement of each BASELINE.json config that is not the headline (driver-visible
    C3 / C4 / C5): same timed region as the headline (K steps between device syncs, median of R; W = 100
    untimed steps first -- a leg starts on a GPU that idled while the host built its filter, and with 10
    warm-up steps the seven regions of round 3 were measured on the clock ramp: C3 read 71 G/s here
    against 78 in a run of its own).
    shrink > 0 (functional tests without a GPU): every N = 2^shrink, C5 with 2 islands."""
    legs = [("c3_systematic", dict(name="c3", scheme="systematic")),
            ("c3_stratified", dict(name="c3", scheme="stratified")),
            ("c3_multinomial", dict(name="c3", scheme="multinomial")),
            ("c4", dict(name="c4")),
            ("c4_dense", dict(name="c4", dense=True)),
            ("c4_collapsed", dict(name="c4", collapsed=True)),
            ("c5", dict(name="c5")),
            ("sqmc", dict(name="c2", qmc=True)),
            # the north star's LITERAL parity contract (strict_ancestors=True: the reference's sequential fp64
            # inverse_cdf, resampling.py:484-509, bit for bit) on the headline configuration and on C3
            ("c2_strict", dict(name="c2", strict=True)),
            ("c3_systematic_strict", dict(name="c3", scheme="systematic", strict=True))]
    out = {}
    for key, kw in legs:
        t_leg = time.perf_counter()
        try:
            if shrink:
                kw = dict(kw, log2N=max(shrink, 11) if kw.get("qmc") else shrink)
            wl = make_workload(T=W + R * K, **kw)
            if shrink and kw["name"] == "c5":
                wl.update(N=1 << shrink, log2N=shrink, Nlabel="2^%d" % shrink, islands=2)
            pf = make_filter(wl)
            dts = time_steps(pf, K, W, R)
            rs_rate = float(np.mean(pf._summ()[0, W:, 4]))
            del pf
            dt = float(np.median(dts))
            units = float(wl["N"]) * wl["islands"] * K
            step_GBs = (16.0 * wl["d"] + 40.0) * units / dt / 1e9
            mv_ms, rs_ms, ns, kernels = kernel_profile(wl, 45 if not shrink else 6, W)
            rf = roofline(wl, step_GBs, mv_ms, rs_ms, ns, kernels)
            out[key] = {"workload": "%s, N=%s, %s resampling, ESSrmin=%g, %d filter(s) per GPU"
                                    % (wl["label"], wl["Nlabel"], wl["scheme"], wl["essrmin"], wl["islands"]),
                        "value": units / dt, "unit": "particle-steps/s", "ms_per_step": 1e3 * dt / K,
                        "steps": K, "warmup": W, "reps": R, "resampled_fraction": rs_rate,
                        "step_frac": rf["step_frac"], "kernel": rf["kernel"], "kernel_ms": rf["kernel_ms"],
                        "bound": rf["bound"], "frac": rf["frac"], "step_kernels": kernels,
                        "per_kernel_ms": {k: v["ms"] for k, v in rf["per_kernel"].items()},
                        "launch_bytes": rf["launch_bytes"],
                        "traffic": None, "traffic_source": None,
                        "leg_seconds": time.perf_counter() - t_leg}
            if not shrink:
                add_profile_fractions(rf, wl)
            for k in ("traffic", "traffic_source", "frac_rocprof", "frac_physical", "rocprof_kernel_us", "launch_flop",
                      "mfma_frac", "mfma_per_16_particles", "limiter"):
                if k in rf:
                    out[key][k] = rf[k]
        except Exception as e:          # one failing leg must not cost the headline its line
            out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def small_and_generic_legs(shrink=0):
    """Driver-visible figures for the three regimes `other_workloads`' big fused legs do not cover:
      c1            BASELINE config C1 -- ToySSM, N = 1000, T = 200, systematic, ONE filter: the whole T-loop is one launch of
                    k_filter_small (one workgroup); this is PMMH's inner call (mcmc.py:445-450: one SMC run per MCMC iteration),
                    so the figure that matters is microseconds per RUN (create + run + read logLt, and run alone);
      c1_islands    1024 such filters as islands of one launch (SMC^2's N_theta x N_x batch, smc_samplers.py:1110-1113);
      generic_model a USER-DEFINED StateSpaceModel (PX0 / PX / PY written with the distributions module, core.py:108-197,
                    state_space_models.py:232-259) at N = 2^20 on the template-method step -- device operators, arrays
                    resident in HBM, Philox draws: the drop-in's non-fused path.
    shrink > 0: emulator sizes (functional tests)."""
    import particles_amd as pa
    from particles_amd import kalman, distributions as dists, resampling as rs
    from particles_amd import state_space_models as ssm
    out = {}
    N1, T1 = (1000, 200) if not shrink else (300, 6)
    y = synthetic_data(T1)
    fk = ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y)
    for key, isl, reps in (("c1", 1, 30 if not shrink else 2), ("c1_islands", 1024 if not shrink else 3, 8 if not shrink else 2)):
        t_leg = time.perf_counter()
        try:
            mk = lambda seed: pa.SMC(fk=fk, N=N1, resampling="systematic", ESSrmin=0.5, collect="off", seed=seed, n_islands=isl)
            w = mk(0); w.run(); _ = w.logLts_islands       # (first launch: code object load)
            desc = _describe(w)
            tot, run = [], []
            for r in range(reps):
                t0 = time.perf_counter()
                pf = mk(1 + r)
                t1 = time.perf_counter()
                pf.run()
                ll = pf.logLts_islands                     # (the run's result on the host: what PMMH / SMC^2 read)
                t2 = time.perf_counter()
                tot.append(t2 - t0); run.append(t2 - t1)
            dt, dr = float(np.median(tot)), float(np.median(run))
            units = float(N1) * T1 * isl
            alg = 56.0 * units                             # SURVEY 8d's accounting (16 d + 40 B per particle-step)
            out[key] = {"workload": "C1: ToySSM d=1 bootstrap filter, N=%d, T=%d, systematic, ESSrmin=0.5, %d filter(s) in one launch"
                                    % (N1, T1, isl),
                        "value": units / dr, "unit": "particle-steps/s", "us_per_run": 1e6 * dr, "us_per_run_with_create": 1e6 * dt,
                        "us_per_step": 1e6 * dr / T1, "filters": isl, "reps": reps, "step_kernels": desc,
                        "logLt_first": float(np.atleast_1d(ll)[0]), "kalman_logLt": kalman_loglik_toy(y),
                        "roofline": {"bound": "hbm", "achieved": alg / dr / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": alg / dr / 1e9 / HBM_PEAK_GBS, "traffic": None, "limiter": "latency+valu",
                                     "note": "one workgroup per filter runs all T steps in ONE launch with the particles in LDS: "
                                             "`achieved` prices SURVEY 8d's 56 B per particle-step, bytes that never reach HBM here "
                                             "(what does: y, the summary ring, the final state); a single filter occupies one CU "
                                             "of 256 and is bound by the T dependent steps, the batch by VALU issue"},
                        "leg_seconds": time.perf_counter() - t_leg}
        except Exception as e:
            out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
    t_leg = time.perf_counter()
    try:
        class UserToySSM(ssm.StateSpaceModel):     # what a user of the reference writes (README.md:66-72)
            default_params = {"sigma": 0.2}

            def PX0(self):
                return dists.Normal()

            def PX(self, t, xp):
                return dists.Normal(loc=xp)

            def PY(self, t, xp, x):
                return dists.Normal(loc=x, scale=self.sigma)
        Ng, K, W = (1 << 20, 20, 3) if not shrink else (1 << shrink, 2, 1)
        yg = synthetic_data(W + K + 1)
        mode, res = _lib_rng_mode(), _lib_resident()
        pa.set_resident(True); rs.set_rng("philox")
        try:
            pf = pa.SMC(fk=ssm.Bootstrap(ssm=UserToySSM(sigma=0.2), data=yg), N=Ng, resampling="systematic", ESSrmin=0.5,
                        collect="off")
            for _ in range(W):
                next(pf)
            pf.sync() if hasattr(pf, "sync") else None
            t0 = time.perf_counter()
            for _ in range(K):
                next(pf)
            ll = float(pf.logLt)                       # (forces the last step's results)
            dt = (time.perf_counter() - t0) / K
            fused = bool(getattr(pf, "_fused", False))
        finally:
            pa.set_resident(res); rs.set_rng(mode)
        if fused:
            raise RuntimeError("the user-defined model took the fused step: the leg would not measure the operator path")
        out["generic_model"] = {"workload": "user-defined StateSpaceModel (ToySSM written with distributions.Normal), Bootstrap, "
                                            "N=%d, systematic, ESSrmin=0.5: template-method step on device operators "
                                            "(set_resident(True), Philox draws)" % Ng,
                                "value": Ng / dt, "unit": "particle-steps/s", "ms_per_step": 1e3 * dt, "steps": K, "warmup": W,
                                "logLt": ll, "kalman_logLt": kalman_loglik_toy(yg[:W + K]),
                                "roofline": {"bound": "hbm", "achieved": 56.0 * Ng / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": 56.0 * Ng / dt / 1e9 / HBM_PEAK_GBS, "traffic": None, "limiter": "host+launches",
                                             "note": "about a dozen operator launches per step, each enqueued from Python through "
                                                     "ctypes, with one host sync for the ESS decision (core.py:327): bound by the "
                                                     "host, not by the device; the fused descriptors exist to avoid exactly this"},
                                "leg_seconds": time.perf_counter() - t_leg}
    except Exception as e:
        out["generic_model"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def _describe(pf):
    import ctypes
    from particles_amd import _lib
    desc = ctypes.create_string_buffer(256)
    _lib.check(_lib.lib().smc_filter_describe(pf._f, desc, 256))
    return desc.value.decode()


def _lib_rng_mode():
    from particles_amd import _lib
    return _lib.RNG_MODE[0]


def _lib_resident():
    from particles_amd import _lib
    return bool(_lib.RESIDENT[0])


def _spawn_ranks(n):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=os.environ.get("MASTER_PORT", str(port)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        kind, N, nsteps, nruns = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
        return _cpu_worker(kind, N, nsteps, nruns, sys.argv[6] if len(sys.argv) > 6 else "",
                           int(sys.argv[7]) if len(sys.argv) > 7 else 0, int(sys.argv[8]) if len(sys.argv) > 8 else 0)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--log2N", type=int, default=None)
    ap.add_argument("--N", type=int, default=0,
                    help="population size that is not a power of two (side measurements; the headline is --log2N 20)")
    ap.add_argument("--scheme", default="systematic")
    ap.add_argument("--islands", type=int, default=1, help="filters per GPU")
    ap.add_argument("--essrmin", type=float, default=None)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json config: c2 = headline (default)")
    ap.add_argument("--cpu-steps", type=int, default=150)
    ap.add_argument("--reps", type=int, default=0,
                    help="repetitions of the K-step timed region (0: about 10^4 timed steps in all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the bounded C3 / C4 / C5 measurements added to the default line")
    ap.add_argument("--other-shrink", type=int, default=0, help=argparse.SUPPRESS)     # tests: other_workloads at 2^k
    ap.add_argument("--collapsed", action="store_true",
                    help="c4: the collapsed form of the optimal proposal's weight (SMC_FLAG_COLLAPSED_PROPOSAL)")
    ap.add_argument("--dense", action="store_true",
                    help="c4: SMC_PATH_MV_DENSE -- dense MFMA products although the model's noise factors are diagonal (the `c4_dense` leg)")
    ap.add_argument("--allow-host-gather", action="store_true",
                    help="N > 1: if RCCL cannot be initialised, gather the evidences over the host rendezvous "
                         "(labelled in the line) instead of failing")
    ap.add_argument("--qmc", action="store_true", help="c2: SMC(qmc=True), the fused SQMC step (the `sqmc` leg)")
    ap.add_argument("--graph", action="store_true", help="replay the steps from hipGraphs (default: eager launches)")
    ap.add_argument("--strict", action="store_true",
                    help="c2 / c3 / c5: strict_ancestors=True -- the reference's sequential fp64 inverse_cdf (resampling.py:484-509), "
                         "literally (the `*_strict` legs)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1 and "RANK" not in os.environ:
            # not under a launcher: spawn the N ranks here (one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
            # MASTER_* as torch.distributed.run would set them -- the library's own TCP rendezvous needs nothing else;
            # the torchrun form keeps working, the 8-GPU line just does not depend on torch being installed)
            return _spawn_ranks(a.gpus)
        a.gpus = world
    # (functional test of the multi-rank path on a box with fewer GPUs than ranks:
    #  SMC_BENCH_NGPU=1 lets all ranks share device 0; RCCL then refuses, which is an ERROR unless
    #  --allow-host-gather routes the evidences over the host rendezvous)
    ngpu = int(os.environ.get("SMC_BENCH_NGPU", "0"))
    device = local_rank % ngpu if ngpu > 0 else local_rank
    os.environ["SMC_HIP_DEVICE"] = str(device)

    from particles_amd import _lib
    from particles_amd.distributed import Group

    # N > 1: the path's one collective is RCCL's.  If RCCL cannot be initialised the run FAILS (rank 0
    # prints the reason as a JSON object with "error" and exits non-zero): a scale line must never be a
    # host-star line mistaken for an xGMI one.  --allow-host-gather (functional tests on a box with fewer
    # GPUs than ranks) opts into the labelled host gather: "rccl": false, "evidence_gather": "host-fallback: ..."
    if a.allow_host_gather:
        os.environ["SMC_ALLOW_HOST_GATHER"] = "1"
    else:
        os.environ.pop("SMC_ALLOW_HOST_GATHER", None)
    grp = None
    if world > 1:
        try:
            grp = Group(device_collective=True)
        except RuntimeError as e:
            if rank == 0:
                print(json.dumps({"error": str(e), "rccl": False, "n_gpus": world,
                                  "hint": "bench.py --allow-host-gather gathers the evidences over the host rendezvous"}),
                      flush=True)
            sys.exit(3)
    if grp and grp.rank == 0 and grp.evidence_path != "rccl":
        print("bench.py: RCCL unavailable, evidences gathered over the host rendezvous (%s)" % grp.evidence_path,
              file=sys.stderr)
    devices = None
    if grp:
        # one GPU per rank: the ranks' PCI bus ids must be distinct unless the caller shares devices
        # on purpose (SMC_BENCH_NGPU: functional tests on a box with fewer GPUs than ranks)
        try:
            pci = _lib.ctx().device_pci()
        except Exception:                   # (a bus id the runtime cannot name must not cost the line)
            pci = ""
        devices = grp.allgather_str("%d:%s" % (device, pci))
        pcis = [s.split(":", 1)[1] for s in devices]
        if ngpu == 0 and all(pcis) and len(set(pcis)) != len(pcis):
            sys.exit("bench.py: ranks share a GPU (%s): launch one rank per GPU" % devices)
    K, W = a.steps, a.warmup
    heavy = a.workload in ("c3", "c4", "c5")        # 0.07-0.3 ms per step: fewer timed steps do
    R = a.reps if a.reps > 0 else max(3, min(500, -(-(2000 if heavy else 10000) // K)))
    T = W + R * K + (K if world > 1 else 0)      # (N > 1: one more K-step region, timed WITH the evidence gather)
    wl = make_workload(a.workload, T, scheme=a.scheme, log2N=a.log2N, N=a.N, islands=a.islands,
                       essrmin=a.essrmin, collapsed=a.collapsed, qmc=a.qmc, strict=a.strict, dense=a.dense)
    a.log2N, a.islands = wl["log2N"], wl["islands"]
    N, d = wl["N"], wl["d"]
    bytes_step = 16.0 * d + 40.0

    pf = make_filter(wl, rank, graph=a.graph)
    # ---- timed region: exactly K steps, barrier + device sync on both sides; R repetitions.
    # Every rank reads its clock right after ITS device sync; the closing barrier follows, and the
    # repetition's time is the MAX over ranks of those local times -- the instant the last rank
    # finished, without the latency of the host-side barrier itself (a TCP star: ~0.1 ms, which at
    # K = 20 steps of 20 us would be a quarter of the region).
    dts = time_steps(pf, K, W, R, grp)
    local_ll = pf.logLts_islands
    # the path's one collective: the all-gather of the per-island evidences, ONCE PER RUN (after the
    # T steps of a filter, not after every K-step repetition) -- timed on its own and reported
    gather_ms = None
    with_gather = None
    all_ll = local_ll
    if grp:
        g = np.zeros(5)
        for i in range(5):
            grp.barrier()
            t0 = time.perf_counter()
            all_ll = grp.gather_evidence(local_ll)
            g[i] = time.perf_counter() - t0
        gather_ms = 1e3 * float(np.median(grp.allreduce_max_host(g)))
        dts = grp.allreduce_max_host(dts)                  # per repetition: the slowest rank
        # one K-step region with the path's collective INSIDE it (the reference's multiSMC returns when the
        # last worker's results have been collected, utils.py:178-186): reported beside the headline
        grp.barrier()
        pf.sync()
        t0 = time.perf_counter()
        pf.step_async(K)
        all_ll = grp.gather_evidence(pf.logLts_islands)
        with_gather = float(grp.allreduce_max_host(time.perf_counter() - t0))
    dt = float(np.median(dts))
    summ0 = pf._summ()[0]
    rs_rate = float(np.mean(summ0[W:, 4]))
    # the self-check a reader can hold the timed kernels to: the evidence estimate after the first n steps of THIS run
    # (the filter the timed region ran on) beside the exact Kalman value of the same n observations; cpu_baseline.logLt
    # (another random stream, the same n when --cpu-steps allows) lands in the same block below
    n_chk = int(min(a.cpu_steps, T) if a.log2N >= 18 else min(T, 2000))        # (= the CPU baseline's sample)
    self_check = None
    if a.workload == "c2" and d == 1 and not a.qmc:
        kal = kalman_loglik_toy(wl["fk"].data[:n_chk])
        self_check = {"steps": n_chk, "gpu_logLt": float(summ0[n_chk - 1, 3]), "kalman_logLt": kal,
                      "gpu_minus_kalman": float(summ0[n_chk - 1, 3]) - kal,
                      "note": "log p(y_0..y_{n-1}) of the first n = steps observations: gpu_logLt from the run the timed region "
                              "belongs to (Philox stream 123), kalman_logLt exact (scalar Kalman filter, bench.kalman_loglik_toy), "
                              "cpu_logLt the CPU baseline's estimate on its own random stream; a particle estimate at N = 2^20 "
                              "has a standard deviation of a few 10^-2 here (profiles/r12g_bias_check.txt)"}
    del pf

    # C5 at the reference's own seam: ONE call of multiSMC (core.py:431-518) by every rank -- the
    # 32 x n_gpus runs sharded over the ranks as islands, the per-run log-evidences gathered by the
    # group's RCCL all-gather -- over K steps of the same data, timed WITH everything the call does
    # (filter construction, the T-loop, the gather): reported beside the headline, whose timed region
    # holds the steps only
    multi = None
    if a.workload == "c5" and a.N == 0:
        import particles_amd as pa
        wl5 = make_workload("c5", K, scheme=a.scheme, essrmin=a.essrmin, log2N=a.log2N, islands=a.islands)
        nruns = wl5["islands"] * world
        kw5 = dict(nruns=nruns, fk=wl5["fk"], N=wl5["N"], resampling=wl5["scheme"], ESSrmin=wl5["essrmin"],
                   group=grp, out_func=lambda pf_: pf_.logLt)
        np.random.seed(4242)
        pa.multiSMC(**kw5)                           # (first call: allocations, RCCL channels)
        if grp:
            grp.barrier()
        t0 = time.perf_counter()
        res = pa.multiSMC(**kw5)
        sec = time.perf_counter() - t0
        sec = float(grp.allreduce_max_host(sec)) if grp else sec
        lls = np.array([r["output"] for r in res])
        from particles_amd.distributed import log_mean_exp_host
        multi = {"call": "particles_amd.multiSMC(nruns=%d, fk=Bootstrap(ToySSM, T=%d), N=2^%d, group=Group(), "
                         "out_func=lambda pf: pf.logLt)" % (nruns, K, wl5["log2N"]),
                 "nruns": nruns, "seconds": sec, "value": nruns * float(wl5["N"]) * K / sec,
                 "unit": "particle-steps/s", "evidence_gather": grp.evidence_path if grp else "none",
                 "log_mean_exp_evidence": log_mean_exp_host(lls), "logLt_sd_over_runs": float(lls.std()),
                 "distinct_runs": int(len(set(lls.tolist())))}

    out = None
    if rank == 0:
        units = float(N) * wl["islands"] * K * world
        out = {
            "metric": "particle-steps/sec (N x T), %s filter N=%s" % ("guided" if wl["guided"] else "bootstrap", wl["Nlabel"]),
            "value": units / dt, "unit": "particle-steps/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": 1e3 * dt / K, "higher_is_better": True,
            "timing": {"reps": R, "statistic": "median of the repetitions of the K-step region",
                       "ms_per_step_min": 1e3 * float(dts.min()) / K,
                       "ms_per_step_max": 1e3 * float(dts.max()) / K,
                       "ms_per_step_p10_p90": [1e3 * float(np.percentile(dts, 10)) / K,
                                               1e3 * float(np.percentile(dts, 90)) / K]},
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s, N=%s, T=%d, %s resampling, ESSrmin=%g; %d independent "
                                   "filter(s) per GPU" % (wl["label"], wl["Nlabel"], K, wl["scheme"], wl["essrmin"],
                                                          wl["islands"]),
                       "N": N, "islands_per_gpu": wl["islands"], "scheme": wl["scheme"],
                       "rng": _lib.lib().smc_version().decode().split("rng=")[-1].split()[0]
                       if b"rng=" in _lib.lib().smc_version() else "philox4x32-10",
                       "graph": bool(a.graph), "resampled_fraction": rs_rate},
            "step_achieved_GBs": bytes_step * N * wl["islands"] * K / dt / 1e9,
            "logLt": [float(v) for v in np.atleast_1d(all_ll)][:16],
            "evidence_gather": grp.evidence_path if grp else "none",
            "evidence_gather_ms": gather_ms,
            # the library's Group raises when RCCL cannot be initialised and bench.py then fails; only
            # --allow-host-gather produces a line with rccl false (= no collective touched xGMI)
            "rccl": bool(grp and grp.evidence_path == "rccl") if grp else None,
        }
        if self_check is not None:
            out["self_check"] = self_check
        if devices is not None:
            out["rank_devices"] = devices
        if multi is not None:
            out["multiSMC"] = multi
        if grp:
            out["timing"]["ms_per_step_with_gather"] = 1e3 * with_gather / K
            out["timing"]["note"] = (
                "per repetition: max over ranks of each rank's own [barrier, device sync, clock] ... K steps ... "
                "[device sync, clock], closing barrier after the clock; the all-gather of the evidences happens once "
                "per run of T steps and is timed separately (evidence_gather_ms = %.3f ms = %.2f %% of a T = 1000 run)"
                % (gather_ms, 100.0 * gather_ms / (1e3 * dt / K * 1000.0)))

    # ---- dominant-kernel duration (HIP events on the filter's stream, outside the timed region)
    if not a.no_profile:
        mv_ms, rs_ms, ns, kernels = kernel_profile(wl, K, W, rank)
        if rank == 0 and ns:
            out["roofline"] = add_profile_fractions(roofline(wl, out["step_achieved_GBs"], mv_ms, rs_ms, ns, kernels), wl)
            out["roofline"]["note"] = ROOFLINE_NOTE
            if a.workload == "c2" and d == 1 and not a.qmc and world == 1:
                # what actually limits this leg (profiles/r13c_size_sweep.txt): the step's time is flat from
                # N = 10^4 to 10^5 -- two dependent launches -- and k_propagate is VALU-bound in its active window; the
                # floor is measured here, on the same filter at N = 2^14
                try:
                    kf, rf_ = (100, 4) if not a.other_shrink else (2, 1)       # (functional tests on the emulator: a token run)
                    wl_f = make_workload("c2", W + kf * rf_, scheme=a.scheme, log2N=14 if not a.other_shrink else 11,
                                         essrmin=a.essrmin, strict=a.strict)
                    pf_f = make_filter(wl_f)
                    floor = float(np.median(time_steps(pf_f, kf, W, rf_))) / kf
                    del pf_f
                    out["roofline"].update({
                        "limiter": "valu+latency", "launch_floor_us": 1e6 * floor,
                        "launch_floor_frac_of_step": floor / (dt / K),
                        "limiter_note": "bound/frac price the dominant kernel against the HBM roofline as the contract asks, but "
                                        "this leg is NOT bandwidth-bound: launch_floor_us is the same step at N = 2^14 (the "
                                        "dependent launches with almost no data), and the kernel's active window is VALU-bound "
                                        "(656 vector instructions per wave of 4 particles; its 46 MB working set lives in the "
                                        "Infinity Cache)"})
                except Exception as e:
                    out["roofline"]["limiter_error"] = "%s: %s" % (type(e).__name__, e)
    if rank == 0 and world == 1 and a.workload == "c2" and a.N == 0 and not a.no_other_workloads \
            and (a.log2N == 20 or a.other_shrink):
        out["other_workloads"] = (other_workloads(K=3, W=2, R=2, shrink=a.other_shrink) if a.other_shrink
                                  else other_workloads())
        out["other_workloads"].update(small_and_generic_legs(shrink=a.other_shrink))
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.workload == "c2":
        nst = min(a.cpu_steps, T) if a.log2N >= 18 else min(T, 2000)
        try:
            out["cpu_baseline"] = cpu_baseline(N, nst, Tdata=T)
            if "self_check" in out and nst == out["self_check"]["steps"]:
                out["self_check"]["cpu_logLt"] = out["cpu_baseline"]["logLt"]
                out["self_check"]["cpu_minus_kalman"] = out["cpu_baseline"]["logLt"] - out["self_check"]["kalman_logLt"]
        except Exception as e:          # a reported baseline, not the measurement: never costs the line
            out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if grp:
        grp.close()


if __name__ == "__main__":
    main()
