#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of the product library against ablation builds (tools/build_ablations.sh).
# $1 = output tag, $2 = space-separated variants ("NEW" = the product library), $3 = bench args
TAG=${1:-ab}; VARS=${2:-"NEW PHILOX_LATE"}; ARGS=${3:-"--steps 1000 --warmup 50 --no-cpu-baseline --no-other-workloads"}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in 1 2 3; do for lib in $VARS; do
  if [ $lib = NEW ]; then unset SMC_HIP_LIBRARY; else export SMC_HIP_LIBRARY=$R/particles_amd/lib/abl/libsmc_$lib.so; fi
  timeout 300 python bench.py $ARGS > $O/ab_${lib}_$rep.json 2>/dev/null
  python - $O/ab_${lib}_$rep.json $lib <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d['roofline']
print(sys.argv[2], 'ms/step %.5f'%d['ms_per_step'], 'G/s %.2f'%(d['value']/1e9), {k[-24:]:round(v['ms'],5) for k,v in r.get('per_kernel',{}).items()})
PY
done; done; unset SMC_HIP_LIBRARY
