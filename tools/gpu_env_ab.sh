#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of code paths selected by environment switches (particles_amd/_lib.py PATH_FLAGS)
# or by alternative library builds.  $1 = tag, $2 = ';'-separated variants, each "name[:VAR=val[,VAR=val]]"
# (name LIB=<path> loads another build), $3 = bench args, $4 = repetitions
TAG=${1:-ab}; VARS=${2:-"base"}; ARGS=${3:-"--steps 1000 --warmup 50 --no-cpu-baseline --no-other-workloads"}; REPS=${4:-3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
IFS=';' read -ra VL <<< "$VARS"
for rep in $(seq 1 $REPS); do for v in "${VL[@]}"; do
  name=${v%%:*}; envs=""; [ "$v" != "$name" ] && envs=${v#*:}
  envcmd=$(echo $envs | tr ',' ' ' | sed 's|LIB=|SMC_HIP_LIBRARY='$R'/particles_amd/lib/abl/|')
  env $envcmd timeout 300 python bench.py $ARGS > $O/ab_${name}_$rep.json 2>/dev/null
  python - $O/ab_${name}_$rep.json $name <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d.get('roofline',{})
    print('%-14s'%sys.argv[2], 'us/step %.3f'%(1e3*d['ms_per_step']), 'G/s %.2f'%(d['value']/1e9), {k[-26:]:round(1e3*v['ms'],3) for k,v in r.get('per_kernel',{}).items()}, 'logLt %.9f'%d['logLt'][0])
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
done; done
