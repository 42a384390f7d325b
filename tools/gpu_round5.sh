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
t)  # in-kernel timelines of the strict step (needs particles_amd/lib/abl/libsmc_TRACE.so)
    python tools/trace_strict.py 20 systematic > $O/r14_trace_strict_c2.txt 2>&1; cat $O/r14_trace_strict_c2.txt
    python tools/trace_strict.py 22 systematic sv > $O/r14_trace_strict_c3.txt 2>&1; cat $O/r14_trace_strict_c3.txt
    ;;
esac
