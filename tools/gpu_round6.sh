#!/bin/bash
# Run ON THE GPU BOX (via gpurun).  $1 = stage, TAG = output tag (default r15)
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r15}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
c4line() {   # $1 = label, rest = env assignments
  local lab=$1; shift
  env "$@" timeout 300 python bench.py --workload c4 $C4ARGS --steps 40 --warmup 10 --reps 5 --no-cpu-baseline --no-other-workloads > $O/c4_$lab.json 2>$O/c4_$lab.err
  python - $O/c4_$lab.json $lab <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d.get('roofline',{})
    print(sys.argv[2], 'us/step %.2f'%(1e3*d['ms_per_step']), 'G/s %.3f'%(d['value']/1e9), 'kernel_ms', r.get('kernel_ms'), 'frac %.3f'%r.get('frac',0), r.get('step_kernels'), 'logLt', d['logLt'][:1])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
case $1 in
micro)  # fp64 MFMA shadow: do fillers between one wave's MFMAs hide?
    ./tools/micro/shadow.bin > $O/${TAG}_mfma_shadow.txt 2>&1; cat $O/${TAG}_mfma_shadow.txt
    ;;
legab)  # same-box A/B of one bench leg: $2 = bench args, $3 = "label:ENV=1,ENV2=2 ..." items
    for rep in 1 2; do
      for item in $3; do
        lab=${item%%:*}; spec=${item#*:}
        env ${spec//,/ } timeout 300 python bench.py $2 --no-cpu-baseline --no-other-workloads > $O/ab_${lab}_$rep.json 2>$O/ab_${lab}_$rep.err
        python - $O/ab_${lab}_$rep.json $lab <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d.get('roofline',{})
    print(sys.argv[2], 'us/step %.2f'%(1e3*d['ms_per_step']), 'G/s %.3f'%(d['value']/1e9), {k[-30:]:round(1e3*v['ms'],2) for k,v in r.get('per_kernel',{}).items()}, r.get('step_kernels'))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
      done
    done
    ;;
c4ab)   # same-box A/B of k_propagate_mv variants: $2 = space separated "label:ENV=1" items
    for rep in 1 2; do
      for item in ${2:-"base:SMC_X=0"}; do
        spec=${item#*:}; c4line ${item%%:*}_$rep ${spec//,/ }
      done
    done
    ;;
c4prof) # trace + counters of the C4 leg
    bash tools/gpu_profile_all.sh $TAG "${2:-c4 c4_collapsed}"
    ;;
tests)
    timeout 1500 python -m pytest tests -m gpu -x -q ${2:+-k "$2"} > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/${TAG}_pytest_gpu.log
    tail -8 $O/${TAG}_pytest_gpu.log
    ;;
full)  # the whole GPU suite, the driver's bench line, K = 1000
    timeout 1500 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/${TAG}_pytest_gpu.log
    tail -6 $O/${TAG}_pytest_gpu.log
    timeout 900 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_line.json 2> $O/${TAG}_bench_driver_line.err; echo "bench rc $?"
    timeout 300 python bench.py --steps 1000 --warmup 50 --no-other-workloads --no-cpu-baseline > $O/${TAG}_bench_c2_k1000.json 2>/dev/null
    python - <<PY
import json
d = json.load(open("$O/${TAG}_bench_driver_line.json"))
print("C2", d["value"] / 1e9, d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("launch_floor_us"), d.get("self_check"))
for k, v in d["other_workloads"].items():
    print(k, v.get("value", 0) / 1e9, v.get("ms_per_step"), v.get("frac"), v.get("error"))
d = json.load(open("$O/${TAG}_bench_c2_k1000.json"))
print("C2 K=1000", d["value"] / 1e9, d["ms_per_step"])
PY
    ;;
line)  # the driver's bench line only
    timeout 900 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_line.json 2> $O/${TAG}_bench_driver_line.err; echo "bench rc $?"
    python - <<PY
import json
d = json.load(open("$O/${TAG}_bench_driver_line.json"))
print("C2", d["value"] / 1e9, d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("launch_floor_us"), d.get("self_check"))
for k, v in d["other_workloads"].items():
    print(k, v.get("value", 0) / 1e9, v.get("ms_per_step"), v.get("frac"), v.get("error"))
PY
    ;;
spacing)
    python tools/trace_spacing.py 22 > $O/${TAG}_trace_spacing.txt 2>&1; cat $O/${TAG}_trace_spacing.txt
    python tools/trace_step.py 22 multinomial sv > $O/${TAG}_trace_c3_multinomial.txt 2>&1; head -12 $O/${TAG}_trace_c3_multinomial.txt
    ;;
sort)   # the radix sort: correctness fuzz, timings, the SQMC leg
    timeout 900 python tools/sort_fuzz.py ${2:-120} > $O/${TAG}_sort_fuzz.txt 2>&1; tail -3 $O/${TAG}_sort_fuzz.txt
    timeout 600 python tools/sort_perf.py > $O/${TAG}_sort_perf.txt 2>&1; cat $O/${TAG}_sort_perf.txt
    ;;
sortab) # the sort's forms on stressing data shapes, the SQMC step on each
    timeout 900 python tools/sort_quick.py > $O/${TAG}_sort_quick.txt 2>&1; cat $O/${TAG}_sort_quick.txt
    ;;
floor)  # in-kernel timelines + the step's floor breakdown (needs particles_amd/lib/abl/libsmc_TRACE.so)
    (python tools/trace_step.py 20; python tools/trace_step.py 14; python tools/trace_step.py 22 systematic sv) > $O/${TAG}_c2_floor.txt 2>&1; cat $O/${TAG}_c2_floor.txt
    ;;
prof)
    bash tools/gpu_profile_all.sh $TAG "$2"
    ;;
final)  # the last tree under load: smoke, in-kernel timelines, fuzzers, soaks, sweeps (each under its own timeout)
    (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "rc=$?" >> $O/${TAG}_smoke.log); tail -3 $O/${TAG}_smoke.log
    (python tools/trace_step.py 20; python tools/trace_step.py 14; python tools/trace_step.py 22 systematic sv) > $O/${TAG}_c2_floor_raw.txt 2>&1; grep -A5 "floor breakdown" $O/${TAG}_c2_floor_raw.txt
    python tools/trace_spacing.py 22 > $O/${TAG}_trace_spacing.txt 2>&1; cat $O/${TAG}_trace_spacing.txt
    (timeout 900 python tools/sort_fuzz.py 120 > $O/${TAG}_sort_fuzz.txt 2>&1; echo "rc=$?" >> $O/${TAG}_sort_fuzz.txt); tail -3 $O/${TAG}_sort_fuzz.txt
    (timeout 600 python tools/sort_perf.py > $O/${TAG}_sort_perf.txt 2>&1); cat $O/${TAG}_sort_perf.txt
    (timeout 900 python tools/strict_soak.py > $O/${TAG}_strict_soak.txt 2>&1; echo "rc=$?" >> $O/${TAG}_strict_soak.txt); tail -12 $O/${TAG}_strict_soak.txt
    (timeout 600 python tools/strict_perf.py > $O/${TAG}_strict_perf.txt 2>&1); tail -12 $O/${TAG}_strict_perf.txt
    (timeout 400 python tools/fuzz_paths.py 300 29 > $O/${TAG}_fuzz_paths.txt 2>&1; echo "rc=$?" >> $O/${TAG}_fuzz_paths.txt); tail -3 $O/${TAG}_fuzz_paths.txt
    (timeout 300 python tools/robustness.py > $O/${TAG}_robustness.txt 2>&1; echo "rc=$?" >> $O/${TAG}_robustness.txt); tail -5 $O/${TAG}_robustness.txt
    (timeout 300 python tools/size_sweep.py > $O/${TAG}_size_sweep.txt 2>&1); cat $O/${TAG}_size_sweep.txt
    (timeout 600 python tools/soak.py > $O/${TAG}_soak.txt 2>&1; echo "rc=$?" >> $O/${TAG}_soak.txt); tail -4 $O/${TAG}_soak.txt
    (timeout 300 python tools/microbench.py > $O/${TAG}_microbench.txt 2>&1); tail -15 $O/${TAG}_microbench.txt
    ;;
esac
