"""Summarise rocprofv3 (rocpd sqlite) outputs into a small text table:
per-kernel call count / average duration from the kernel trace, and per-kernel
per-dispatch averages of every PMC counter collected.

    python tools/summarise_prof.py gpurun_out/prof_<tag> [--config JSON] [--command STR] [--summary NAME]

With PMC passes of FETCH_SIZE / WRITE_SIZE present it also writes <dir>/traffic.json =
{"kernels": {name: {..., "hbm_bytes_per_launch"}}, "config": {...}, "summary": NAME, "command": STR}:
the record bench.py's `measured_traffic` matches against the workload it is running (config:
workload key, log2N, islands, scheme), so the three optional arguments are what makes the file usable
as profiles/traffic_<key>.json (tests/test_bench.py::test_committed_traffic_files_are_usable).
"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

import argparse
_ap = argparse.ArgumentParser()
_ap.add_argument("out")
_ap.add_argument("--config", default="")
_ap.add_argument("--command", default="")
_ap.add_argument("--summary", default="")
_a = _ap.parse_args()
out = _a.out

for f in sorted(glob.glob(os.path.join(out, "**", "*_results.db"), recursive=True)):
    c = sqlite3.connect(f)
    rel = os.path.relpath(f, out)
    try:
        rows = list(c.execute("select name, total_calls, total_duration, average, percentage "
                              "from top_kernels"))
    except sqlite3.Error:
        rows = []
    if rows and "trace" in rel:
        print("== kernel trace stats:", rel)
        for name, calls, tot, avg, pct in rows:
            print("  %-44s calls %6d  avg %9.3f us  total %10.3f us  %5.1f%%"
                  % (name[:44], calls, avg, tot, pct))
    try:
        acc = defaultdict(float)
        cnt = defaultdict(int)
        dur = defaultdict(float)
        for k, cn, v, d in c.execute("select kernel_name, counter_name, value, duration "
                                     "from counters_collection"):
            acc[(k[:44], cn)] += v
            cnt[(k[:44], cn)] += 1
            dur[(k[:44], cn)] += d
    except sqlite3.Error:
        acc = {}
    if acc:
        print("== counters (per-dispatch average):", rel)
        for (k, cn), v in sorted(acc.items()):
            if k.startswith("__amd"):
                continue
            n = cnt[(k, cn)]
            print("  %-44s %-22s %18.1f   (n=%d, avg kernel %8.2f us under PMC)"
                  % (k, cn, v / n, n, dur[(k, cn)] / n / 1e3))

# ---- machine-readable HBM traffic per launch (for bench.py's roofline.traffic) and the kernels' average duration
# in the --kernel-trace --stats pass (for roofline.frac_rocprof / frac_physical: the fractions a reader can recompute)
import json
traffic = {}
avg_us = {}
for f in sorted(glob.glob(os.path.join(out, "**", "*_results.db"), recursive=True)):
    if "trace" not in os.path.relpath(f, out):
        continue
    c = sqlite3.connect(f)
    try:
        for name, calls, avg in c.execute("select name, total_calls, average from top_kernels"):
            avg_us[name] = (float(avg), int(calls))
    except sqlite3.Error:
        pass
for f in sorted(glob.glob(os.path.join(out, "**", "*_results.db"), recursive=True)):
    c = sqlite3.connect(f)
    try:
        for k, cn, v, n in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                     "where counter_name in ('FETCH_SIZE','WRITE_SIZE') group by 1,2"):
            traffic.setdefault(k, {})[cn] = v
            traffic[k]["launches"] = n
    except sqlite3.Error:
        pass
def _base(name):
    return name.replace("void ", "").split("(")[0].strip()


def _avg(k):
    """(average us, launches) of kernel k in the trace pass: by its full name, else by name + template arguments."""
    if k in avg_us:
        return avg_us[k]
    hits = [v for n, v in avg_us.items() if _base(n) == _base(k)]
    return hits[0] if len(hits) == 1 else (None, None)


if traffic:
    # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B
    # (MI355X_MICROARCH.md, HBM): double it for wide coalesced reads
    res = {k: {"FETCH_SIZE_KiB": d.get("FETCH_SIZE"), "WRITE_SIZE_KiB": d.get("WRITE_SIZE"),
               "hbm_bytes_per_launch": (2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024.0,
               "launches": d.get("launches"),
               # average duration in the --kernel-trace --stats pass (no counters armed), and its launches
               "avg_us": _avg(k)[0], "trace_launches": _avg(k)[1]}
           for k, d in traffic.items() if not k.startswith("__amd")}
    with open(os.path.join(out, "traffic.json"), "w") as fh:
        rec = {"kernels": res}
        if _a.config:
            rec["config"] = json.loads(_a.config)
        if _a.summary:
            rec["summary"] = _a.summary
        if _a.command:
            rec["command"] = _a.command
        # which tree the record is of: the commit the caller names (SMC_TREE_STAMP: `git rev-parse --short HEAD` plus
        # "-dirty" -- the GPU box has no .git) and a hash of the device sources, which bench.py / the tests recompute
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from source_hash import source_hash
        rec["source_hash"] = source_hash()
        rec["tree"] = os.environ.get("SMC_TREE_STAMP", "unknown")
        json.dump(rec, fh, indent=1)
    print("== traffic.json:", json.dumps(res))
