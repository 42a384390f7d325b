#!/bin/bash
# Run ON THE GPU BOX (via gpurun).  $1 = stage
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r14
mkdir -p $O
cd $R
case $1 in
a)  # strict v2: parity at full size, timings, a kernel trace of the strict C2 step
    timeout 900 python -m pytest tests -m gpu -x -q -k "strict or sequential_prefix" > $O/r14a_pytest_strict.log 2>&1; echo "pytest rc $?" >> $O/r14a_pytest_strict.log
    tail -15 $O/r14a_pytest_strict.log
    timeout 600 python tools/strict_perf.py > $O/r14a_strict_perf.txt 2>&1; tail -40 $O/r14a_strict_perf.txt
    cd /tmp && export TMPDIR=/tmp
    timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace_c2_strict -o trace -- python $R/bench.py --workload c2 --strict --steps 200 --warmup 20 --reps 3 --no-cpu-baseline --no-other-workloads --no-profile > $O/r14a_c2_strict_trace.log 2>&1
    python $R/tools/summarise_prof.py $O/trace_c2_strict > $O/r14a_c2_strict_summary.txt 2>&1; head -20 $O/r14a_c2_strict_summary.txt
    tail -2 $O/r14a_c2_strict_trace.log
    rm -rf $O/trace_c2_strict
    ;;
full)  # the whole GPU suite, the driver's bench line, K = 1000, the profiles of every leg
    timeout 900 python -m pytest tests -m gpu -x -q > $O/${TAG:-r14}_pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/${TAG:-r14}_pytest_gpu.log
    tail -6 $O/${TAG:-r14}_pytest_gpu.log
    timeout 600 python bench.py --steps 20 --warmup 5 > $O/${TAG:-r14}_bench_driver_line.json 2> $O/${TAG:-r14}_bench_driver_line.err; echo "bench rc $?"
    timeout 300 python bench.py --steps 1000 --warmup 50 --no-other-workloads --no-cpu-baseline > $O/${TAG:-r14}_bench_c2_k1000.json 2>/dev/null
    python - <<PY
import json
d = json.load(open("$O/${TAG:-r14}_bench_driver_line.json"))
print("C2", d["value"] / 1e9, d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("launch_floor_us"), d.get("self_check"))
for k, v in d["other_workloads"].items():
    print(k, v.get("value", 0) / 1e9, v.get("ms_per_step"), v.get("frac"), v.get("error"))
d = json.load(open("$O/${TAG:-r14}_bench_c2_k1000.json"))
print("C2 K=1000", d["value"] / 1e9, d["ms_per_step"])
PY
    ;;
prof)
    bash tools/gpu_profile_all.sh ${TAG:-r14} "$2"
    ;;
t)  # in-kernel timelines of the strict step (needs particles_amd/lib/abl/libsmc_TRACE.so)
    python tools/trace_strict.py 20 systematic > $O/r14_trace_strict_c2.txt 2>&1; cat $O/r14_trace_strict_c2.txt
    python tools/trace_strict.py 22 systematic sv > $O/r14_trace_strict_c3.txt 2>&1; cat $O/r14_trace_strict_c3.txt
    ;;
esac
