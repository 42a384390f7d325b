#!/bin/bash
# on the GPU box: time the bench under each ablation library
R=$GRAFT_REPO_ROOT
for v in "" ${ABLS:-NO_RNG NO_MFMA NO_LOAD NO_STORE}; do
  if [ -z "$v" ]; then lib=$R/particles_amd/lib/libsmc_hip.so; else lib=$R/particles_amd/lib/abl/libsmc_$v.so; fi
  SMC_HIP_LIBRARY=$lib timeout 120 python $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline $EXTRA 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline', {})
        print('%-10s ms/step %.4f  propagate %.4f ms  resampling kernels %.4f ms' % ('${v:-base}', d['ms_per_step'], r.get('kernel_ms', 0), r.get('prepare_ms', 0)))
    elif 'rror' in l: print(l.strip())
"
done
