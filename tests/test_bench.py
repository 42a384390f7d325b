"""bench.py's contract on the CPU (fiber emulator standing in for the GPU): the one JSON line, the
`other_workloads` block that makes C3 / C4 / C5 driver-visible, and the CPU-baseline legs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config")


def _emu_env():
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    return dict(os.environ, SMC_TEST_EMULATOR="1", SMC_HIP_LIBRARY=build_emu.build())


def test_default_line_carries_other_workloads_and_cpu_legs(tmp_path):
    """`python bench.py` (N = 1, default workload) shrunk to emulator sizes: top-level fields are C2's,
    `other_workloads` holds one measurement per remaining BASELINE.json config, `cpu_baseline` states
    the host's core count and has a one-core and an all-cores leg."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--log2N", "11",
           "--reps", "2", "--other-shrink", "10", "--cpu-steps", "4"]
    p = subprocess.run(cmd, env=_emu_env(), capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and "C2" in d["config"]["workload"]
    assert "roofline" in d and d["roofline"]["bound"] == "hbm"
    ow = d["other_workloads"]
    assert sorted(ow) == ["c1", "c1_islands", "c2_strict", "c3_multinomial", "c3_stratified", "c3_systematic",
                          "c3_systematic_strict", "c4", "c4_collapsed", "c4_dense", "c5", "generic_model", "sqmc"]
    # BASELINE config C1 (one small filter: the PMMH regime), its batched form, and a user-defined model on the operator path
    small = {k: ow.pop(k) for k in ("c1", "c1_islands", "generic_model")}
    for key, leg in small.items():
        assert "error" not in leg, (key, leg)
        assert leg["value"] > 0 and leg["roofline"]["bound"] == "hbm" and "limiter" in leg["roofline"], (key, leg)
    assert "k_filter_small" in small["c1"]["step_kernels"] and small["c1"]["filters"] == 1 and small["c1"]["us_per_run"] > 0
    assert small["c1_islands"]["filters"] > 1 and abs(small["c1"]["logLt_first"] - small["c1"]["kalman_logLt"]) < 3.0
    assert "user-defined" in small["generic_model"]["workload"]
    # the literal-parity contract (strict_ancestors) is on the driver's line, with its own kernels
    for key in ("c2_strict", "c3_systematic_strict"):
        assert "k_strict_classify+k_strict_search" in ow[key]["step_kernels"] or "k_sqx_classify" in ow[key]["step_kernels"], ow[key]
        assert "strict_ancestors" in ow[key]["workload"]
    # what limits C2, and the self-check a reader can hold the timed kernels to
    assert d["roofline"]["limiter"] == "valu+latency" and d["roofline"]["launch_floor_us"] >= 0
    sc = d["self_check"]
    assert sc["steps"] == 7 and abs(sc["gpu_logLt"] - sc["kalman_logLt"]) < 1.0 and abs(sc["cpu_logLt"] - sc["kalman_logLt"]) < 1.0
    for k in ("frac", "frac_rocprof", "frac_physical"):
        assert k in d["roofline"], k
    assert "k_sq_permute" in ow["sqmc"]["step_kernels"] and "k_rs_sort" in ow["sqmc"]["step_kernels"]
    for key, leg in ow.items():
        assert "error" not in leg, (key, leg)
        for k in ("value", "ms_per_step", "step_frac", "kernel", "frac", "step_kernels"):
            assert k in leg, (key, k)
        assert leg["value"] > 0 and leg["resampled_fraction"] > 0
    # C4's model has diagonal noise factors: the element-wise form (HBM is then the bounding roofline); `c4_dense` keeps
    # the dense MFMA products measurable
    assert ow["c4"]["bound"] == "hbm" and "k_propagate_mv" in ow["c4"]["step_kernels"] and "[diagonal factors]" in ow["c4"]["step_kernels"]
    assert ow["c4"]["mfma_per_16_particles"] == 32 and ow["c4"]["limiter"] == "valu"
    assert ow["c4_dense"]["bound"] == "mfma" and "[diagonal factors]" not in ow["c4_dense"]["step_kernels"]
    assert "collapsed" in ow["c4_collapsed"]["step_kernels"]
    assert "k_f_spacing" in ow["c3_multinomial"]["step_kernels"] or "spacing" in ow["c3_multinomial"]["step_kernels"]
    cb = d["cpu_baseline"]
    assert cb["cores"] == 1 and cb["value"] > 0 and cb["host"]["nproc"] >= 1
    assert cb["kind"] == ("reference" if os.path.isdir("/root/reference/particles") else "port")
    if cb["host"]["nproc"] > 1:
        ac = cb["all_cores"]
        assert ac["cores"] == min(64, cb["host"]["nproc"]) and ac["host_nproc"] == cb["host"]["nproc"]
        assert ac["runs"] >= 16 and ac["value"] > 0 and ac["kind"] == cb["kind"]
        # independent runs, as multiSMC's are (utils.py:189-213): every worker its own seeds (64 copies of one)
        assert ac["distinct_logLt"] == ac["runs"] and ac["logLt_sd"] > 0


@pytest.mark.parametrize("kind", ["port", "reference"])
def test_cpu_baseline_kinds(kind):
    """Both CPU legs on demand: the oracle port (what the GPU box runs) and -- where
    /root/reference exists -- the reference itself; the same data and seeds, so the two
    log-evidences agree to the last bit (the oracle is pinned to the reference)."""
    import bench
    if kind == "reference" and not bench.reference_available():
        pytest.skip("/root/reference is not present on this box")
    cb = bench.cpu_baseline(1 << 12, 6, all_cores=False, kind=kind)
    assert cb["kind"] == kind and cb["cores"] == 1 and cb["value"] > 0
    if bench.reference_available():
        other = bench.cpu_baseline(1 << 12, 6, all_cores=False, kind="port" if kind == "reference" else "reference")
        assert other["logLt"] == cb["logLt"]


def test_auto_kind_follows_the_reference_tree(monkeypatch):
    import bench
    monkeypatch.setattr(bench, "REFERENCE_DIR", "/nonexistent")
    assert not bench.reference_available()
    assert bench.cpu_baseline(1 << 10, 4, all_cores=False)["kind"] == "port"


def test_committed_traffic_files_are_usable():
    """Every committed profiles/traffic_<key>.json must be a record bench.py can use: written by
    tools/summarise_prof.py WITH its config / summary / command, naming a summary that is committed
    beside it, and `measured_traffic` must return a number for the workload the record describes
    (round 3's driver line carried `roofline.traffic: null` because a refreshed file had lost its
    `config`)."""
    import glob
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_*.json")))
    assert files
    for f in files:
        rec = json.load(open(f))
        key = os.path.basename(f)[len("traffic_"):-len(".json")]
        for k in ("kernels", "config", "summary", "command"):
            assert k in rec, (f, k)
        assert os.path.exists(os.path.join(ROOT, "profiles", rec["summary"])), (f, rec["summary"])
        cfg = rec["config"]
        wl = bench.make_workload(cfg["workload"], 4, scheme=cfg["scheme"], collapsed=bool(cfg.get("collapsed")),
                                 qmc=bool(cfg.get("qmc")), strict=bool(cfg.get("strict")), dense=bool(cfg.get("dense")))
        assert bench.leg_key(wl) == key, (f, bench.leg_key(wl))
        assert (wl["log2N"], wl["islands"]) == (cfg["log2N"], cfg["islands"]), f
        names = [n for n in rec["kernels"] if "k_propagate" in n]
        assert names, f
        tr = bench.measured_traffic(wl, "k_propagate_mv" if cfg["workload"] == "c4" else "k_propagate")
        assert tr is not None and tr[0] > 1e6, (f, tr)
        # algorithmic bytes of the propagate launch against what the counters saw: within 2x
        alg = (16.0 * wl["d"] + 24.0) * wl["N"] * wl["islands"]
        assert 0.4 < tr[0] / alg < 2.0, (f, tr[0] / alg)
        # a different size is NOT this record's workload
        wl2 = bench.make_workload(cfg["workload"], 4, scheme=cfg["scheme"], log2N=cfg["log2N"] - 1,
                                  collapsed=bool(cfg.get("collapsed")), qmc=bool(cfg.get("qmc")), strict=bool(cfg.get("strict")),
                                  dense=bool(cfg.get("dense")))
        if cfg["workload"] != "c5":
            assert bench.measured_traffic(wl2, "k_propagate") is None


def test_fractions_are_reproducible_from_profiles():
    """every leg's `frac_rocprof` and `frac_physical` must be what a reader computes by hand from the
    committed record -- algorithmic (or counted) bytes / the kernel's average duration in the rocprofv3 trace pass / the
    peak -- and that duration must be the one printed in the summary file the record names."""
    import glob
    import re
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_*.json")))
    assert len(files) >= 11
    for f in files:
        rec = json.load(open(f))
        cfg = rec["config"]
        wl = bench.make_workload(cfg["workload"], 4, scheme=cfg["scheme"], collapsed=bool(cfg.get("collapsed")),
                                 qmc=bool(cfg.get("qmc")), strict=bool(cfg.get("strict")), dense=bool(cfg.get("dense")))
        # the record is of THIS tree's kernels (tools/source_hash.py: csrc/*, the C header, the compiler flags) -- a leg
        # whose kernels changed after its record was taken must be profiled again before the tree is committed
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from source_hash import source_hash
        assert rec.get("source_hash") == source_hash(), (f, rec.get("source_hash"), source_hash(), rec.get("tree"))
        assert rec.get("tree", "unknown") != "unknown", f
        mv = "k_propagate_mv" if cfg["workload"] == "c4" else "k_propagate"
        name, d = [(n, v) for n, v in rec["kernels"].items() if bench_base(n) == mv][0]
        assert d["avg_us"] and d["avg_us"] > 1.0, f
        # the duration in the record is the one in the summary text (the trace pass's table)
        txt = open(os.path.join(ROOT, "profiles", rec["summary"])).read()
        rows = re.findall(r"^\s+(.*?)\s+calls\s+\d+\s+avg\s+([0-9.]+) us", txt, flags=re.M)
        avgs = [float(a) for n_, a in rows if n_.replace("void ", "").startswith(mv + ("<" if "<" in name else "("))
                or n_.replace("void ", "").split("<")[0].split("(")[0] == mv]
        assert any(abs(a - d["avg_us"]) < 5e-4 for a in avgs), (f, avgs, d["avg_us"])
        N, isl, dd = wl["N"], wl["islands"], wl["d"]
        two = any(k in " ".join(rec["kernels"]) for k in ("k_ancestors2", "k_strict_classify", "k_strict_step"))
        rf = {"kernel": mv, "bound": "hbm", "launch_bytes": (16.0 * dd + 16.0 + (8.0 if two else 0.0)) * N * isl}
        if cfg["workload"] == "c4" and cfg.get("dense"):   # (the dense MFMA products: priced against the fp64 matrix peak)
            rf.update(bound="mfma", launch_flop=bench.mfma_per_16(bool(cfg.get("collapsed")), False) * 2048.0 / 16.0 * N * isl)
        bench.add_profile_fractions(rf, wl)
        sec = d["avg_us"] * 1e-6
        if rf["bound"] == "mfma":
            assert abs(rf["frac_rocprof"] - rf["launch_flop"] / sec / 1e12 / bench.FP64_PEAK_TF) < 1e-12
            assert 0.3 < rf["frac_rocprof"] < 0.8, (f, rf["frac_rocprof"])
        else:
            assert abs(rf["frac_rocprof"] - rf["launch_bytes"] / sec / 1e9 / bench.HBM_PEAK_GBS) < 1e-12
            assert abs(rf["frac_physical"] - d["hbm_bytes_per_launch"] / sec / 1e9 / bench.HBM_PEAK_GBS) < 1e-12
            # (the counters see fewer bytes than SURVEY's accounting -- 32-bit ancestors -- except under SQMC, whose
            #  k_propagate also reads the tape of ndtri values)
            assert 0.25 < rf["frac_physical"] < 1.0 and 0.3 < rf["frac_rocprof"] < 1.0, (f, rf)
            assert cfg.get("qmc") or rf["frac_physical"] <= rf["frac_rocprof"] * 1.05, (f, rf)


def bench_base(name):
    return name.replace("void ", "").split("<")[0].split("(")[0].strip()
