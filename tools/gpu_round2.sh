#!/bin/bash
# Run ON THE GPU BOX (gpurun): round-2 measurement batch.  $1 = tag, $2 = what ("all" | list)
TAG=${1:-r02a}
WHAT=${2:-"pytest bench ab c345 prof balance"}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
jl() { grep '^{' "$1" | tail -1; }
for w in $WHAT; do case $w in
pytest)
  (timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log)
  tail -5 $O/pytest_gpu.log ;;
smoke)
  (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log); tail -3 $O/smoke.log ;;
bench)
  timeout 300 python bench.py --steps 1000 --warmup 50 > $O/bench_c2_k1000.json 2> $O/bench_c2_k1000.err
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2_k20.json 2> $O/bench_c2_k20.err
  timeout 300 python bench.py > $O/bench_c2_default.json 2> $O/bench_c2_default.err
  for f in $O/bench_c2_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d.get('roofline',{})
print(sys.argv[1].split('/')[-1], 'ms/step %.5f'%d['ms_per_step'], 'G/s %.2f'%(d['value']/1e9), d.get('timing'), {k:round(v['ms'],5) for k,v in r.get('per_kernel',{}).items()}, 'cpu', d.get('cpu_baseline',{}).get('value'))
PY
  done ;;
ab)
  SMC_ANC2_R1=1 timeout 300 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > $O/bench_c2_anc2r1.json 2>&1
  SMC_NO_NT=1 timeout 300 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > $O/bench_c2_nont.json 2>&1
  timeout 300 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --graph > $O/bench_c2_graph.json 2>&1
  for f in $O/bench_c2_anc2r1.json $O/bench_c2_nont.json $O/bench_c2_graph.json; do python - "$f" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d.get('roofline',{})
print(sys.argv[1].split('/')[-1], 'ms/step %.5f'%d['ms_per_step'], 'G/s %.2f'%(d['value']/1e9), {k:round(v['ms'],5) for k,v in r.get('per_kernel',{}).items()})
PY
  done ;;
c345)
  timeout 300 python bench.py --workload c5 --steps 200 --warmup 20 > $O/bench_c5.json 2>&1
  for sc in systematic stratified multinomial; do timeout 300 python bench.py --workload c3 --scheme $sc --steps 200 --warmup 20 > $O/bench_c3_$sc.json 2>&1; done
  timeout 300 python bench.py --workload c4 --steps 100 --warmup 10 > $O/bench_c4.json 2>&1
  for f in $O/bench_c5.json $O/bench_c3_*.json $O/bench_c4.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], 'ms/step %.5f'%d['ms_per_step'], 'G/s %.2f'%(d['value']/1e9), {k:round(v['ms'],5) for k,v in r.get('per_kernel',{}).items()})
except Exception as e: print(sys.argv[1], 'FAILED', e, open(sys.argv[1]).read()[-400:])
PY
  done ;;
prof)
  EXTRA="" bash tools/gpu_profile.sh ${TAG}_c2 400 > $O/prof_c2.txt 2>&1; tail -40 $O/prof_c2.txt ;;
prof_r1)
  export SMC_ANC2_R1=1; PASSES="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" bash tools/gpu_profile.sh ${TAG}_c2_anc2r1 400 > $O/prof_c2_r1.txt 2>&1; unset SMC_ANC2_R1; tail -25 $O/prof_c2_r1.txt ;;
prof_c5)
  EXTRA="--workload c5" bash tools/gpu_profile.sh ${TAG}_c5 100 > $O/prof_c5.txt 2>&1; tail -30 $O/prof_c5.txt ;;
benchN)
  for n in 100000 1000000 10000000; do timeout 300 python bench.py --N $n --steps 400 --warmup 50 --no-cpu-baseline > $O/bench_N$n.json 2>&1; done
  for sc in stratified multinomial; do timeout 300 python bench.py --N 1000000 --scheme $sc --essrmin 1.0 --steps 400 --warmup 50 --no-cpu-baseline > $O/bench_N1000000_$sc.json 2>&1; done
  for f in $O/bench_N*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], 'ms/step %.5f'%d['ms_per_step'], 'G/s %.2f'%(d['value']/1e9), {k[-28:]:round(v['ms'],5) for k,v in r.get('per_kernel',{}).items()})
except Exception as e: print(sys.argv[1], 'FAILED', e, open(sys.argv[1]).read()[-400:])
PY
  done ;;
abold)
  # same box, the r03b-state library against the current one (C2, K = 1000 and K = 20, twice each, interleaved)
  for rep in 1 2; do for lib in R03B NEW; do
    if [ $lib = R03B ]; then export SMC_HIP_LIBRARY=$R/particles_amd/lib/abl/libsmc_R03B.so; else unset SMC_HIP_LIBRARY; fi
    timeout 300 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > $O/ab_${lib}_k1000_$rep.json 2>&1
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/ab_${lib}_k20_$rep.json 2>&1
    timeout 300 python bench.py --workload c5 --steps 200 --warmup 20 --no-cpu-baseline > $O/ab_${lib}_c5_$rep.json 2>&1
  done; done; unset SMC_HIP_LIBRARY
  for f in $O/ab_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], 'ms/step %.5f'%d['ms_per_step'], 'G/s %.2f'%(d['value']/1e9), {k[-20:]:round(v['ms'],5) for k,v in r.get('per_kernel',{}).items()})
except Exception as e: print(sys.argv[1], 'FAILED', e, open(sys.argv[1]).read()[-300:])
PY
  done ;;
examples)
  for e in examples/*.py; do echo "== $e"; (timeout 200 python $e 2>&1 | tail -6); done > $O/examples.txt 2>&1
  echo "== sharded smc2, 2 ranks on this GPU (host gather opt-in)" >> $O/examples.txt
  (SMC_ALLOW_HOST_GATHER=1 SMC_HIP_DEVICE=0 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29677 examples/smc2_toy.py 2>&1 | grep -v "^\*\|OMP_NUM" | tail -5) >> $O/examples.txt 2>&1
  cat $O/examples.txt ;;
robust)
  (timeout 200 python tools/robustness.py > $O/robustness.txt 2>&1; echo "rc=$?" >> $O/robustness.txt); cat $O/robustness.txt
  (timeout 200 python tools/soak.py > $O/soak.txt 2>&1; echo "rc=$?" >> $O/soak.txt); cat $O/soak.txt
  (timeout 300 python tools/fuzz_paths.py 400 23 > $O/fuzz400.txt 2>&1; echo "rc=$?" >> $O/fuzz400.txt); tail -3 $O/fuzz400.txt ;;
sweep)
  timeout 300 python tools/size_sweep.py > $O/size_sweep.txt 2>&1; cat $O/size_sweep.txt ;;
fuzz)
  timeout 900 python tools/fuzz_paths.py 150 11 > $O/fuzz.txt 2>&1; tail -4 $O/fuzz.txt ;;
smc2prof)
  timeout 300 python tools/smc2_profile.py > $O/smc2_profile.txt 2>&1; cat $O/smc2_profile.txt ;;
quick)
  (timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log); tail -3 $O/pytest_gpu.log
  timeout 300 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > $O/bench_c2_k1000.json 2>&1
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2_k20.json 2>&1
  timeout 300 python bench.py --workload c5 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c5.json 2>&1
  for sc in systematic stratified multinomial; do timeout 300 python bench.py --workload c3 --scheme $sc --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c3_$sc.json 2>&1; done
  for f in $O/bench_c2_k1000.json $O/bench_c2_k20.json $O/bench_c5.json $O/bench_c3_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], 'ms/step %.5f'%d['ms_per_step'], 'G/s %.2f'%(d['value']/1e9), {k[-28:]:round(v['ms'],5) for k,v in r.get('per_kernel',{}).items()})
except Exception as e: print(sys.argv[1], 'FAILED', e, open(sys.argv[1]).read()[-400:])
PY
  done ;;
c3m)
  (timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log); tail -3 $O/pytest_gpu.log
  SMC_FLAT_MULTINOMIAL=1 timeout 300 python bench.py --workload c3 --scheme multinomial --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c3_multinomial_flat.json 2>&1
  timeout 300 python bench.py --workload c3 --scheme multinomial --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c3_multinomial.json 2>&1
  python - "$O/bench_c3_multinomial.json" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d.get('roofline',{})
print('c3 multinomial ms/step %.5f'%d['ms_per_step'], 'G/s %.2f'%(d['value']/1e9), {k:round(v['ms'],5) for k,v in r.get('per_kernel',{}).items()})
d=json.loads([l for l in open(sys.argv[1].replace('.json','_flat.json')) if l.startswith('{')][-1]); r=d.get('roofline',{})
print('   (flat path) ms/step %.5f'%d['ms_per_step'], 'G/s %.2f'%(d['value']/1e9), {k:round(v['ms'],5) for k,v in r.get('per_kernel',{}).items()})
PY
  ;;
prof_benchN)
  for n in 100000 1000000 10000000; do timeout 300 python bench.py --N $n --steps 400 --warmup 50 --no-cpu-baseline > $O/bench_N$n.json 2>&1; done
  for sc in stratified multinomial; do timeout 300 python bench.py --N 1000000 --scheme $sc --essrmin 1.0 --steps 400 --warmup 50 --no-cpu-baseline > $O/bench_N1000000_$sc.json 2>&1; done
  for f in $O/bench_N*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], 'ms/step %.5f'%d['ms_per_step'], 'G/s %.2f'%(d['value']/1e9), {k[-28:]:round(v['ms'],5) for k,v in r.get('per_kernel',{}).items()})
except Exception as e: print(sys.argv[1], 'FAILED', e, open(sys.argv[1]).read()[-400:])
PY
  done ;;
abold)
  # same box, the r03b-state library against the current one (C2, K = 1000 and K = 20, twice each, interleaved)
  for rep in 1 2; do for lib in R03B NEW; do
    if [ $lib = R03B ]; then export SMC_HIP_LIBRARY=$R/particles_amd/lib/abl/libsmc_R03B.so; else unset SMC_HIP_LIBRARY; fi
    timeout 300 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > $O/ab_${lib}_k1000_$rep.json 2>&1
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/ab_${lib}_k20_$rep.json 2>&1
    timeout 300 python bench.py --workload c5 --steps 200 --warmup 20 --no-cpu-baseline > $O/ab_${lib}_c5_$rep.json 2>&1
  done; done; unset SMC_HIP_LIBRARY
  for f in $O/ab_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], 'ms/step %.5f'%d['ms_per_step'], 'G/s %.2f'%(d['value']/1e9), {k[-20:]:round(v['ms'],5) for k,v in r.get('per_kernel',{}).items()})
except Exception as e: print(sys.argv[1], 'FAILED', e, open(sys.argv[1]).read()[-300:])
PY
  done ;;
examples)
  for e in examples/*.py; do echo "== $e"; (timeout 200 python $e 2>&1 | tail -6); done > $O/examples.txt 2>&1
  echo "== sharded smc2, 2 ranks on this GPU (host gather opt-in)" >> $O/examples.txt
  (SMC_ALLOW_HOST_GATHER=1 SMC_HIP_DEVICE=0 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29677 examples/smc2_toy.py 2>&1 | grep -v "^\*\|OMP_NUM" | tail -5) >> $O/examples.txt 2>&1
  cat $O/examples.txt ;;
robust)
  (timeout 200 python tools/robustness.py > $O/robustness.txt 2>&1; echo "rc=$?" >> $O/robustness.txt); cat $O/robustness.txt
  (timeout 200 python tools/soak.py > $O/soak.txt 2>&1; echo "rc=$?" >> $O/soak.txt); cat $O/soak.txt
  (timeout 300 python tools/fuzz_paths.py 400 23 > $O/fuzz400.txt 2>&1; echo "rc=$?" >> $O/fuzz400.txt); tail -3 $O/fuzz400.txt ;;
sweep)
  timeout 300 python tools/size_sweep.py > $O/size_sweep.txt 2>&1; cat $O/size_sweep.txt ;;
fuzz)
  timeout 900 python tools/fuzz_paths.py 150 11 > $O/fuzz.txt 2>&1; tail -4 $O/fuzz.txt ;;
smc2prof)
  timeout 300 python tools/smc2_profile.py > $O/smc2_profile.txt 2>&1; cat $O/smc2_profile.txt ;;
quick)
  (timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log); tail -3 $O/pytest_gpu.log
  timeout 300 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > $O/bench_c2_k1000.json 2>&1
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2_k20.json 2>&1
  timeout 300 python bench.py --workload c5 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c5.json 2>&1
  for sc in systematic stratified multinomial; do timeout 300 python bench.py --workload c3 --scheme $sc --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c3_$sc.json 2>&1; done
  for f in $O/bench_c2_k1000.json $O/bench_c2_k20.json $O/bench_c5.json $O/bench_c3_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], 'ms/step %.5f'%d['ms_per_step'], 'G/s %.2f'%(d['value']/1e9), {k[-28:]:round(v['ms'],5) for k,v in r.get('per_kernel',{}).items()})
except Exception as e: print(sys.argv[1], 'FAILED', e, open(sys.argv[1]).read()[-400:])
PY
  done ;;
c3m)
  EXTRA="--workload c3 --scheme multinomial" PASSES="FETCH_SIZE;WRITE_SIZE" bash tools/gpu_profile.sh ${TAG}_c3m 60 > $O/prof_c3m.txt 2>&1; tail -30 $O/prof_c3m.txt ;;
trace)
  timeout 200 python tools/trace_step.py 20 > $O/trace_step.txt 2>&1; cat $O/trace_step.txt ;;
abtk)
  for v in 0 1; do
    if [ $v = 1 ]; then export SMC_NO_TK=1; else unset SMC_NO_TK; fi
    timeout 300 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > $O/bench_c2_notk$v.json 2>&1
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2k20_notk$v.json 2>&1
    timeout 300 python bench.py --workload c5 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c5_notk$v.json 2>&1
    timeout 300 python bench.py --workload c3 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c3_notk$v.json 2>&1
  done; unset SMC_NO_TK
  for f in $O/bench_*notk*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], 'ms/step %.5f'%d['ms_per_step'], 'G/s %.2f'%(d['value']/1e9), {k:round(v['ms'],5) for k,v in r.get('per_kernel',{}).items()})
except Exception as e: print(sys.argv[1], 'FAILED', e, open(sys.argv[1]).read()[-400:])
PY
  done ;;
bench2)
  # the N = 2 launch line of the driver, both ranks on this box's one GPU (functional: RCCL may refuse 2 ranks per device)
  SMC_BENCH_NGPU=1 SMC_ALLOW_HOST_GATHER=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $O/bench_2ranks_1gpu.json 2> $O/bench_2ranks_1gpu.err
  python - "$O/bench_2ranks_1gpu.json" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print('2 ranks / 1 GPU: ms/step %.5f'%d['ms_per_step'], 'G/s %.2f'%(d['value']/1e9), d['evidence_gather'], 'gather ms', d['evidence_gather_ms'], d['timing'].get('note','')[-90:])
except Exception as e: print('bench2 FAILED', e, open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
  ;;
sqmc)
  timeout 200 python tools/sort_perf.py > $O/sort_perf.txt 2>&1; cat $O/sort_perf.txt ;;
balance)
  timeout 120 python tools/tile_balance.py 20 > $O/tile_balance.txt 2>&1; cat $O/tile_balance.txt ;;
esac; done
