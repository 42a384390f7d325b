#!/bin/bash
# Run ON THE GPU BOX (gpurun): round-4 evidence on the last tree.  $1 = tag (files land in gpurun_out/<tag>/)
TAG=${1:-r13c}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], 'us/step %.3f'%(1e3*d['ms_per_step']), 'G/s %.2f'%(d['value']/1e9),
          {k[-24:]:round(1e3*v['ms'],3) for k,v in r.get('per_kernel',{}).items()}, 'step_frac %.3f'%r.get('step_frac',0), 'frac %.3f'%r.get('frac',0), 'traffic', r.get('traffic'))
    for k,v in d.get('other_workloads',{}).items():
        print('   ', k, v.get('error') or ('%.2f G/s  %.1f us/step  step_frac %.3f  %s frac %.3f traffic %s' % (v['value']/1e9, 1e3*v['ms_per_step'], v['step_frac'], v['kernel'][-28:], v['frac'], v.get('traffic'))))
    c=d.get('cpu_baseline')
    if c: print('    cpu', c['kind'], '1 core %.1f M/s'%(c['value']/1e6), 'nproc', c['host']['nproc'], 'all cores %.1f M/s'%(c.get('all_cores',{}).get('value',0)/1e6))
    if 'multiSMC' in d: print('    multiSMC', d['multiSMC'])
except Exception as e: print(sys.argv[1], 'FAILED', e, open(sys.argv[1]).read()[-600:])
PY
}
(timeout 2400 python -m pytest tests -m gpu -x -q --durations=10 > $O/${TAG}_pytest_gpu.log 2>&1; echo "rc=$?" >> $O/${TAG}_pytest_gpu.log); tail -4 $O/${TAG}_pytest_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "rc=$?" >> $O/${TAG}_smoke.log); tail -3 $O/${TAG}_smoke.log
SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver_line.json 2> $O/${TAG}_bench_driver_line.err; echo "driver line wall ${SECONDS}s"
line $O/${TAG}_bench_driver_line.json; tail -3 $O/${TAG}_bench_driver_line.err
timeout 300 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-other-workloads > $O/${TAG}_bench_c2_k1000.json 2>/dev/null; line $O/${TAG}_bench_c2_k1000.json
for sc in systematic stratified multinomial; do timeout 300 python bench.py --workload c3 --scheme $sc --steps 100 --warmup 10 > $O/${TAG}_bench_c3_$sc.json 2>/dev/null; line $O/${TAG}_bench_c3_$sc.json; done
timeout 300 python bench.py --workload c5 --steps 200 --warmup 20 > $O/${TAG}_bench_c5.json 2>/dev/null; line $O/${TAG}_bench_c5.json
timeout 300 python bench.py --workload c4 --steps 100 --warmup 10 > $O/${TAG}_bench_c4.json 2>/dev/null; line $O/${TAG}_bench_c4.json
timeout 300 python bench.py --workload c4 --collapsed --steps 100 --warmup 10 > $O/${TAG}_bench_c4_collapsed.json 2>/dev/null; line $O/${TAG}_bench_c4_collapsed.json
(timeout 300 python tools/strict_perf.py > $O/${TAG}_strict_perf.txt 2>&1); tail -4 $O/${TAG}_strict_perf.txt
(timeout 300 python tools/size_sweep.py > $O/${TAG}_size_sweep.txt 2>&1); cat $O/${TAG}_size_sweep.txt
(timeout 400 python tools/fuzz_paths.py 300 29 > $O/${TAG}_fuzz_paths.txt 2>&1; echo "rc=$?" >> $O/${TAG}_fuzz_paths.txt); tail -3 $O/${TAG}_fuzz_paths.txt
(timeout 300 python tools/robustness.py > $O/${TAG}_robustness.txt 2>&1; echo "rc=$?" >> $O/${TAG}_robustness.txt); tail -5 $O/${TAG}_robustness.txt
# two ranks on this box's one GPU: RCCL refuses -> the launch line FAILS with its JSON error; with --allow-host-gather the labelled line
SMC_BENCH_NGPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-profile > $O/${TAG}_bench_2ranks_1gpu_default.json 2> /dev/null; echo "2 ranks, default: rc=$? $(head -c 300 $O/${TAG}_bench_2ranks_1gpu_default.json)"
SMC_BENCH_NGPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 20 --warmup 5 --workload c5 --no-cpu-baseline --no-profile --allow-host-gather > $O/${TAG}_bench_2ranks_1gpu_c5.json 2> /dev/null; line $O/${TAG}_bench_2ranks_1gpu_c5.json
bash tools/gpu_profile_all.sh $TAG > $O/${TAG}_profile_all.log 2>&1; tail -3 $O/${TAG}_profile_all.log
