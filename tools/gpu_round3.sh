#!/bin/bash
# Run ON THE GPU BOX (gpurun): round-3 measurement batches.  $1 = tag, $2 = stages
TAG=${1:-r04a}
WHAT=${2:-"rates pytest smoke bench abrng"}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], 'ms/step %.5f'%d['ms_per_step'], 'G/s %.2f'%(d['value']/1e9),
          {k[-24:]:round(v['ms'],5) for k,v in r.get('per_kernel',{}).items()}, 'step_frac %.3f'%r.get('step_frac',0))
    for k,v in d.get('other_workloads',{}).items():
        print('   ', k, v.get('error') or ('%.2f G/s  %.4f ms/step  step_frac %.3f  %s frac %.3f' % (v['value']/1e9, v['ms_per_step'], v['step_frac'], v['kernel'][-20:], v['frac'])))
    c=d.get('cpu_baseline')
    if c: print('    cpu', c['kind'], '1 core %.1f M/s'%(c['value']/1e6), 'nproc', c['host']['nproc'], 'all cores %.1f M/s'%(c.get('all_cores',{}).get('value',0)/1e6))
except Exception as e: print(sys.argv[1], 'FAILED', e, open(sys.argv[1]).read()[-600:])
PY
}
for w in $WHAT; do case $w in
rates)
  (timeout 300 tools/micro/_build/rates > $O/rates.txt 2>&1; echo "rc=$?" >> $O/rates.txt); cat $O/rates.txt ;;
pytest)
  (timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log)
  tail -25 $O/pytest_gpu.log ;;
smoke)
  (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log); tail -3 $O/smoke.log ;;
bench)
  SECONDS=0; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench_driver_line.err; echo "driver line wall ${SECONDS}s"
  line $O/bench_driver_line.json; tail -3 $O/bench_driver_line.err
  timeout 300 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-other-workloads > $O/bench_c2_k1000.json 2> $O/bench_c2_k1000.err
  line $O/bench_c2_k1000.json ;;
abrng)
  # one box: table-free Box-Muller (rounds 1-2) | table-driven (this tree) | table-driven + Philox4x32-7
  for rep in 1 2; do for lib in BM_LEGACY NEW PHILOX_ROUNDS=7; do
    if [ $lib = NEW ]; then unset SMC_HIP_LIBRARY; else export SMC_HIP_LIBRARY=$R/particles_amd/lib/abl/libsmc_$lib.so; fi
    timeout 300 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-other-workloads > $O/ab_${lib}_c2k1000_$rep.json 2>&1
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads > $O/ab_${lib}_c2k20_$rep.json 2>&1
    timeout 300 python bench.py --workload c5 --steps 100 --warmup 10 > $O/ab_${lib}_c5_$rep.json 2>&1
    if [ $rep = 1 ]; then
      timeout 300 python bench.py --workload c4 --steps 50 --warmup 10 > $O/ab_${lib}_c4_$rep.json 2>&1
      timeout 300 python bench.py --workload c4 --collapsed --steps 50 --warmup 10 > $O/ab_${lib}_c4coll_$rep.json 2>&1
      timeout 300 python bench.py --workload c3 --steps 100 --warmup 10 > $O/ab_${lib}_c3_$rep.json 2>&1
    fi
  done; done; unset SMC_HIP_LIBRARY
  for f in $O/ab_*.json; do line $f; done ;;
prof)
  EXTRA="--no-cpu-baseline --no-other-workloads" bash tools/gpu_profile.sh ${TAG}_c2 400 > $O/prof_c2.txt 2>&1; tail -40 $O/prof_c2.txt ;;
sqmc)
  (timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sqmc" > $O/pytest_sqmc.log 2>&1; echo "rc=$?" >> $O/pytest_sqmc.log); tail -8 $O/pytest_sqmc.log
  timeout 300 python tools/sqmc_perf.py > $O/sqmc_perf.txt 2>&1; cat $O/sqmc_perf.txt ;;
prof_sqmc)
  P=$O/prof_sqmc; mkdir -p $P
  (cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $P/trace -o trace -- python $R/tools/sqmc_perf.py 20 30 > $P/trace.log 2>&1)
  python tools/summarise_prof.py $P > $O/sqmc_summary.txt 2>&1; rm -rf $P/trace; find $P -name "*.db" -delete 2>/dev/null
  tail -5 $P/trace.log; head -40 $O/sqmc_summary.txt
  timeout 300 python tools/sqmc_perf.py > $O/sqmc_perf.txt 2>&1; cat $O/sqmc_perf.txt ;;
prof_c3m)
  EXTRA="--workload c3 --scheme multinomial" bash tools/gpu_profile.sh ${TAG}_c3m 100 > $O/prof_c3m.txt 2>&1; tail -40 $O/prof_c3m.txt ;;
c3m)
  for sc in multinomial stratified systematic; do
    timeout 300 python bench.py --workload c3 --scheme $sc --steps 100 --warmup 10 > $O/bench_c3_$sc.json 2>&1; line $O/bench_c3_$sc.json; done ;;
robust)
  (timeout 300 python tools/robustness.py > $O/robustness.txt 2>&1; echo "rc=$?" >> $O/robustness.txt); tail -25 $O/robustness.txt
  (timeout 300 python tools/soak.py > $O/soak.txt 2>&1; echo "rc=$?" >> $O/soak.txt); tail -8 $O/soak.txt
  (timeout 400 python tools/fuzz_paths.py 400 23 > $O/fuzz400.txt 2>&1; echo "rc=$?" >> $O/fuzz400.txt); tail -3 $O/fuzz400.txt ;;
sweep)
  (timeout 300 python tools/size_sweep.py > $O/size_sweep.txt 2>&1); cat $O/size_sweep.txt ;;
trace)
  (timeout 200 python tools/trace_step.py > $O/trace_step.txt 2>&1); cat $O/trace_step.txt ;;
trace_c3m)
  (timeout 200 python tools/trace_step.py 22 multinomial sv > $O/trace_c3m.txt 2>&1); cat $O/trace_c3m.txt
  (timeout 200 python tools/trace_step.py 22 systematic sv > $O/trace_c3s.txt 2>&1); cat $O/trace_c3s.txt ;;
prof_c5)
  EXTRA="--workload c5" bash tools/gpu_profile.sh ${TAG}_c5 100 > $O/prof_c5.txt 2>&1; tail -30 $O/prof_c5.txt ;;
prof_c4)
  EXTRA="--workload c4" bash tools/gpu_profile.sh ${TAG}_c4 50 > $O/prof_c4.txt 2>&1; tail -24 $O/prof_c4.txt ;;
c45)
  timeout 300 python bench.py --workload c5 --steps 200 --warmup 20 > $O/bench_c5.json 2>&1; line $O/bench_c5.json
  timeout 300 python bench.py --workload c4 --steps 100 --warmup 10 > $O/bench_c4.json 2>&1; line $O/bench_c4.json
  timeout 300 python bench.py --workload c4 --collapsed --steps 100 --warmup 10 > $O/bench_c4_collapsed.json 2>&1; line $O/bench_c4_collapsed.json ;;
two_procs_c3m)
  # two processes sharing the one GPU, each a C3 multinomial filter (the merged spacings + reduction launch waits
  # on lower-numbered workgroups only: progress must not depend on every workgroup being resident)
  (timeout 300 python bench.py --workload c3 --scheme multinomial --steps 100 --warmup 10 --no-profile > $O/c3m_proc_a.json 2>&1 &
   timeout 300 python bench.py --workload c3 --scheme multinomial --steps 100 --warmup 10 --no-profile > $O/c3m_proc_b.json 2>&1; wait)
  for f in $O/c3m_proc_a.json $O/c3m_proc_b.json; do python - $f <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[1].split('/')[-1], 'ms/step %.5f'%d['ms_per_step'], 'G/s %.2f'%(d['value']/1e9), 'logLt', d['logLt'][:1])
PY
  done ;;
two_ranks)
  # the driver's N = 2 launch line with both ranks on this box's one GPU (RCCL refuses: labelled host fallback)
  SMC_BENCH_NGPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks_1gpu.json 2> $O/bench_2ranks_1gpu.err; line $O/bench_2ranks_1gpu.json; tail -2 $O/bench_2ranks_1gpu.err ;;
*)
  echo "unknown stage $w" ;;
esac; done
