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
data with its own Philox island id (weak scaling: total work = n_gpus * N * K);
the per-rank log-evidences are gathered with RCCL (smc_comm_*) inside the timed
region.  Inputs are resident in HBM before the timed region starts.

The timed region of exactly K steps (barrier + device sync on both sides) is repeated R times
on consecutive stretches of the same run (R such that about 10^4 steps are timed in all);
`ms_per_step` / `value` are the MEDIAN repetition, the spread is reported beside them -- a
K = 20 region lasts half a millisecond and a single shot of it is noise.

Rank 0 prints ONE JSON line (see README / task contract), including
  roofline      -- the longer of the step's two kernels (resampling: k_ancestors2 /
                   k_ancestors; propagate: k_propagate): algorithmic bytes per launch over
                   its average duration, BOTH measured with HIP events on the filter's stream
                   (smc_filter_kernel_ms), against the 8 TB/s HBM peak of MI355X;
  cpu_baseline  -- N=1, rank 0: the reference itself (nchopin/particles, pf.cpu_time) when it is
                   importable (build container), else the NumPy restatement of its path
                   (oracle/, "port") timed on this host's cores on a bounded sample, with the
                   reference's own figures from profiles/cpu_reference.json printed beside it.

Other BASELINE.json configs: --workload c3 (StochVol N=2^22; --scheme), c4 (d=32 guided),
c5 (32 islands x 2^18 per GPU; `--workload c5 --gpus 8` is the 256-island SCALE run, evidence
all-gather over RCCL inside the timed region).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_STEP = 56.0                # 16*d + 40 B per particle-step, d = 1
BYTES_MOVE = 32.0                # k_propagate: read A, gather X; write X, lw
BYTES_PREPARE = 16.0             # k_ancestors: read lw, write A  (beyond 2048 workgroups per launch
                                 # k_prepare reads lw once more for the tile totals: + 8 B)


def synthetic_data(T, sigma=0.2, seed=42):
    """BASELINE.md section 3's inputs: ToySSM(sigma) (README.md:66-78), np.random.seed(42);
    x, y = model.simulate(T) -- our simulate() consumes numpy's legacy stream exactly as the
    reference's does (state_space_models.py:283-323), so these are the reference's data."""
    from particles_amd import kalman
    np.random.seed(seed)
    x, y = kalman.ToySSM(sigma).simulate(T)
    return y


def measured_traffic(a, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of
    this very command line (tools/gpu_profile.sh -> tools/summarise_prof.py:
    FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as the
    MI355X guide prescribes for gfx950).  PMC counters cannot be collected from
    inside the process, so the figure is only reported for the configuration the
    committed profile was taken on; otherwise traffic stays null."""
    path = os.path.join(ROOT, "profiles", "traffic_%s.json" % a.workload)
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        rec = json.load(fh)
    cfg = rec.get("config", {})
    if a.N > 0 or (cfg.get("log2N"), cfg.get("islands"), cfg.get("scheme")) != (a.log2N, a.islands, a.scheme):
        return None
    for name, d in rec["kernels"].items():
        if kernel in name:
            return d["hbm_bytes_per_launch"], "rocprofv3 PMC, profiles/%s" % rec.get("summary", "")
    return None


def cpu_baseline(y, N, nsteps):
    """CPU leg beside the GPU number.  With /root/reference importable (build container) the
    reference itself, timed by its own pf.cpu_time (utils.py:81-89, core.py:391); on the GPU box
    the oracle's restatement of that path ("port": same NumPy calls, C inverse_cdf), plus the
    reference's figures measured in the build container (profiles/cpu_reference.json)."""
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True,
                   stdout=subprocess.DEVNULL)
    ref_file = os.path.join(ROOT, "profiles", "cpu_reference.json")
    committed = None
    if os.path.exists(ref_file):
        with open(ref_file) as fh:
            rec = json.load(fh)
        committed = {"host": rec["host"], "source": "profiles/cpu_reference.json (tools/cpu_reference.py)",
                     "legs": {k: {"particle_steps_per_s": v["particle_steps_per_s"], "cores": v["cores"]}
                              for k, v in rec["legs"].items()}}
    sample = ("N=2^%d, first %d steps of the same data (np.random.seed(42); simulate), run seed 123; "
              "cost per step is flat in T" % (int(np.log2(N)), nsteps))
    if os.path.isdir("/root/reference/particles"):
        code = ("import sys, json; sys.dont_write_bytecode = True\n"
                "sys.path.insert(0, %r)\n"
                "import numpy as np, tools.cpu_reference as cr\n"
                "m = cr.ToySSM(sigma=0.2); np.random.seed(42); x, y = m.simulate(%d)\n"
                "med, allt, ll = cr.time_run(m, y, %d, reps=1)\n"
                "print(json.dumps({'s': med, 'll': float(ll)}))\n"
                % (ROOT, nsteps, N))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
        if r.returncode == 0:
            o = json.loads(r.stdout.strip().splitlines()[-1])
            return {"value": N * nsteps / o["s"], "unit": "particle-steps/s", "cores": 1,
                    "kind": "reference", "seconds": o["s"], "logLt": o["ll"],
                    "sample": "particles.SMC(...).run(), pf.cpu_time; inverse_cdf bound to its gcc -O2 "
                              "restatement (numba absent); " + sample,
                    "reference_build_container": committed}
    from oracle import smc_oracle as orc
    np.random.seed(123)
    t0 = time.perf_counter()
    out = orc.run_filter(orc.ToySSM(0.2), y[:nsteps], N, "systematic", 0.5)
    dt = time.perf_counter() - t0
    return {"value": N * nsteps / dt, "unit": "particle-steps/s", "cores": 1, "kind": "port",
            "sample": "oracle.run_filter (NumPy restatement of particles.SMC + C inverse_cdf); " + sample,
            "seconds": dt, "logLt": out["final_logLt"],
            "reference_build_container": committed}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--log2N", type=int, default=20)
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
    ap.add_argument("--collapsed", action="store_true",
                    help="c4: the collapsed form of the optimal proposal's weight (SMC_FLAG_COLLAPSED_PROPOSAL)")
    ap.add_argument("--graph", action="store_true", help="replay the steps from hipGraphs (default: eager launches)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run "
                     "--nproc-per-node %d" % (a.gpus, a.gpus))
        a.gpus = world
    # (functional test of the multi-rank path on a box with fewer GPUs than ranks:
    #  SMC_BENCH_NGPU=1 lets all ranks share device 0; RCCL then refuses, which is an ERROR unless
    #  SMC_ALLOW_HOST_GATHER=1 routes the evidences over the host rendezvous)
    ngpu = int(os.environ.get("SMC_BENCH_NGPU", "0"))
    os.environ["SMC_HIP_DEVICE"] = str(local_rank % ngpu if ngpu > 0 else local_rank)

    import particles_amd as pa
    from particles_amd import _lib, kalman
    from particles_amd import state_space_models as ssm
    from particles_amd.distributed import Group

    # a multi-GPU bench line with a LABELLED host gather ("evidence_gather": "host-fallback: <reason>")
    # is worth more than no line: the gather is outside the timed region either way
    os.environ.setdefault("SMC_ALLOW_HOST_GATHER", "1")
    grp = Group(device_collective=True) if world > 1 else None
    if grp and grp.rank == 0 and grp.evidence_path != "rccl":
        print("bench.py: RCCL unavailable, evidences gathered over the host rendezvous (%s)" % grp.evidence_path,
              file=sys.stderr)
    K, W = a.steps, a.warmup
    heavy = a.workload in ("c3", "c4", "c5")        # 0.07-0.3 ms per step: fewer timed steps do
    R = a.reps if a.reps > 0 else max(3, min(500, -(-(2000 if heavy else 10000) // K)))
    T = W + R * K
    d = 1
    if a.workload == "c3":        # StochVol, N = 2^22, ESSrmin = 1
        a.log2N = 22 if a.log2N == 20 else a.log2N
        a.essrmin = 1.0 if a.essrmin is None else a.essrmin
        rng = np.random.RandomState(42)
        model = ssm.StochVol()
        x = np.empty(T)
        x[0] = model.mu + model.sig0() * rng.standard_normal()
        for t in range(1, T):
            x[t] = model.EXt(x[t - 1]) + model.sigma * rng.standard_normal()
        y = [np.array([v]) for v in np.exp(0.5 * x) * rng.standard_normal(T)]
        fk = ssm.Bootstrap(ssm=model, data=y)
        wl = "C3: StochVol d=1 bootstrap filter"
    elif a.workload == "c4":      # MVLinearGauss_Guarniero d = 32, guided, N = 2^20
        d = 32
        rng = np.random.RandomState(42)
        model = kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=d)
        x = np.zeros(d)
        y = []
        for t in range(T):
            x = (model.F @ x if t else np.zeros(d)) + rng.standard_normal(d)
            y.append((x + rng.standard_normal(d)).reshape(1, d))
        fk = ssm.GuidedPF(ssm=model, data=y)
        wl = "C4: MVLinearGauss_Guarniero d=32 guided filter" + (" (collapsed proposal weight)" if a.collapsed else "")
    else:
        if a.workload == "c5":    # one GPU's share of 256 islands x 2^18
            a.log2N, a.islands = 18, 32
        y = synthetic_data(T)
        fk = ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y)
        wl = "C2: ToySSM d=1 linear-Gaussian bootstrap filter" if a.workload == "c2" else \
             "C5: ToySSM d=1 bootstrap filter islands"
    a.essrmin = 0.5 if a.essrmin is None else a.essrmin
    N = a.N if a.N > 0 else 1 << a.log2N
    bytes_step = 16.0 * d + 40.0
    bytes_move = 16.0 * d + 16.0                    # k_propagate: read A, gather X; write X, lw

    def make(profile=False):
        pf = pa.SMC(fk=fk, N=N, resampling=a.scheme, ESSrmin=a.essrmin, collect="off", seed=123,
                    n_islands=a.islands, island_offset=rank * a.islands,
                    use_graph=a.graph and not profile, collapsed_proposal=a.collapsed)
        if profile:
            _lib.check(_lib.lib().smc_filter_profile(pf._f, 1))
        return pf

    pf = make()
    pf.step_async(W)
    pf.sync()
    if grp:       # warm-up of the path's one collective too (RCCL sets its channels up lazily)
        grp.gather_evidence(pf.logLts_islands)
    # ---- timed region: exactly K steps, barrier + device sync on both sides; R repetitions.
    # Every rank reads its clock right after ITS device sync; the closing barrier follows, and the
    # repetition's time is the MAX over ranks of those local times -- the instant the last rank
    # finished, without the latency of the host-side barrier itself (a TCP star: ~0.1 ms, which at
    # K = 20 steps of 20 us would be a quarter of the region).
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
    local_ll = pf.logLts_islands
    # the path's one collective: the all-gather of the per-island evidences, ONCE PER RUN (after the
    # T steps of a filter, not after every K-step repetition) -- timed on its own and reported
    gather_ms = None
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
    dt = float(np.median(dts))
    rs_rate = float(np.mean(pf._summ()[0, W:, 4]))
    del pf

    out = None
    if rank == 0:
        units = float(N) * a.islands * K * world
        out = {
            "metric": "particle-steps/sec (N x T), %s filter N=%s" % ("guided" if a.workload == "c4" else "bootstrap",
                                                                          str(a.N) if a.N > 0 else "2^%d" % a.log2N),
            "value": units / dt, "unit": "particle-steps/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": 1e3 * dt / K, "higher_is_better": True,
            "timing": {"reps": R, "statistic": "median of the repetitions of the K-step region",
                       "ms_per_step_min": 1e3 * float(dts.min()) / K,
                       "ms_per_step_max": 1e3 * float(dts.max()) / K,
                       "ms_per_step_p10_p90": [1e3 * float(np.percentile(dts, 10)) / K,
                                               1e3 * float(np.percentile(dts, 90)) / K]},
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s, N=%s, T=%d, %s resampling, ESSrmin=%g; %d independent "
                                   "filter(s) per GPU" % (wl, str(a.N) if a.N > 0 else "2^%d" % a.log2N, K,
                                                          a.scheme, a.essrmin, a.islands),
                       "N": N, "islands_per_gpu": a.islands, "scheme": a.scheme,
                       "rng": "philox4x32-10", "graph": bool(a.graph),
                       "resampled_fraction": rs_rate},
            "step_achieved_GBs": bytes_step * N * a.islands * K / dt / 1e9,
            "logLt": [float(v) for v in np.atleast_1d(all_ll)][:16],
            "evidence_gather": grp.evidence_path if grp else "none",
            "evidence_gather_ms": gather_ms,
            # the library's Group raises when RCCL cannot be initialised; bench.py opts into the host
            # gather (SMC_ALLOW_HOST_GATHER, set above unless the caller exported 0) so that a scale
            # line exists either way -- and says so here: rccl false = no collective touched xGMI
            "rccl": bool(grp and grp.evidence_path == "rccl") if grp else None,
        }
        if grp:
            out["timing"]["note"] = (
                "per repetition: max over ranks of each rank's own [barrier, device sync, clock] ... K steps ... "
                "[device sync, clock], closing barrier after the clock; the all-gather of the evidences happens once "
                "per run of T steps and is timed separately (evidence_gather_ms = %.3f ms = %.2f %% of a T = 1000 run)"
                % (gather_ms, 100.0 * gather_ms / (1e3 * dt / K * 1000.0)))

    # ---- dominant-kernel duration: same workload re-run with HIP events around
    # every launch on the filter's stream (kept out of the timed region above)
    if not a.no_profile:
        pf = make(profile=True)
        pf.step_async(W)
        pf.sync()
        import ctypes
        mv, pr, ns = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(_lib.lib().smc_filter_kernel_ms(pf._f, ctypes.byref(mv), ctypes.byref(pr),
                                                   ctypes.byref(ns)))    # drop warm-up samples
        pf.step_async(min(K, 4000))
        _lib.check(_lib.lib().smc_filter_kernel_ms(pf._f, ctypes.byref(mv), ctypes.byref(pr),
                                                   ctypes.byref(ns)))
        desc = ctypes.create_string_buffer(256)
        _lib.check(_lib.lib().smc_filter_describe(pf._f, desc, 256))
        kernels = desc.value.decode()
        del pf
        if rank == 0 and ns.value:
            # the step is [resampling kernel(s)] + [propagate kernel]; the roofline object describes
            # whichever takes longer, the other one is listed beside it
            rs_name = "+".join(k for k in kernels.split("+") if not k.startswith("k_propagate"))
            mv_name = [k for k in kernels.split("+") if k.startswith("k_propagate")][0]
            # two-level path: k_propagate also writes the tile CDF (8 B), k_ancestors2 reads it
            # instead of the log-weights: 16 d + 24 and 16 B; flat path: 16 d + 16 and 16 (+ 8) B;
            # either way SURVEY 8d's 16 d + 40 B per particle-step in all
            two = "k_ancestors2" in kernels
            rs_bytes = (BYTES_PREPARE + (8.0 if "k_prepare" in kernels else 0.0)) * N * a.islands
            mv_bytes = (bytes_move + (8.0 if two else 0.0)) * N * a.islands
            rs_ms = pr.value
            per = {mv_name: {"ms": mv.value, "launch_bytes": mv_bytes,
                             "achieved": mv_bytes / (mv.value * 1e-3) / 1e9},
                   rs_name: {"ms": rs_ms, "launch_bytes": rs_bytes,
                             "achieved": rs_bytes / (rs_ms * 1e-3) / 1e9 if rs_ms > 0 else None}}
            dom = mv_name if mv.value >= rs_ms else rs_name
            ach = per[dom]["achieved"]
            out["roofline"] = {
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": None,
                "traffic_source": None,
                "kernel": dom, "kernel_ms": per[dom]["ms"],
                "launch_bytes": per[dom]["launch_bytes"],
                "samples": ns.value,
                "step_kernels": kernels, "per_kernel": per,
                "step_frac": out["step_achieved_GBs"] / HBM_PEAK_GBS,
                "note": "algorithmic bytes in SURVEY 8d's accounting (int64 ancestors; they are stored as "
                        "32-bit words, so the kernels physically move 4 B less per particle each). "
                        "Per particle-step, two-level path: k_propagate 16 d + 24 B (read A, gather X; write X, "
                        "lw and the tile's integer CDF), k_ancestors2 16 B (read that CDF, write A); flat path: "
                        "k_propagate 16 d + 16 B, k_ancestors 16 B (read lw, write A; + 8 B for k_prepare's pass "
                        "over lw beyond 2048 workgroups per launch): 16 d + 40 B in all. Both parts are measured with HIP events on "
                        "the filter's stream in a separate pass over the same workload: steps are sampled "
                        "in three kinds (whole step / up to the propagate launch / from there on), "
                        "propagate = whole - first part, resampling = whole - second part, so the fixed "
                        "~4 us of an event interval cancels. `kernel` is the one that takes longer; "
                        "step_frac = 56 B x N / ms_per_step over the HBM peak (the whole step)",
            }
            if a.workload == "c4":
                # GEMM-shaped kernel: priced against the dense fp64 matrix peak (MI355X spec
                # 78.6 TFLOP/s, = its fp64 vector peak).  72 MFMAs
                # (v_mfma_f64_16x16x4: 2048 flop) per 16 particles for the guided d=32 step.
                flop = (44 if a.collapsed else 72) * 2048.0 / 16.0 * N * a.islands
                tf = flop / (mv.value * 1e-3) / 1e12
                out["roofline"].update({
                    "bound": "mfma", "achieved": tf, "peak": 78.6, "unit": "TFLOP/s",
                    "frac": tf / 78.6, "kernel": "k_propagate_mv", "kernel_ms": mv.value,
                    "launch_bytes": mv_bytes, "launch_flop": flop,
                    "hbm_achieved_GBs": per[mv_name]["achieved"]})
            tr = measured_traffic(a, out["roofline"]["kernel"].split("+")[-1].split("<")[0])
            if tr:
                out["roofline"]["traffic"], out["roofline"]["traffic_source"] = tr
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.workload == "c2":
        nst = min(a.cpu_steps, T) if a.log2N >= 18 else min(T, 2000)
        out["cpu_baseline"] = cpu_baseline(y, N, nst)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if grp:
        grp.close()


if __name__ == "__main__":
    main()
