cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r12b; mkdir -p $O
B="python $R/bench.py --steps 400 --warmup 50 --reps 3 --no-cpu-baseline --no-other-workloads --no-profile"
for v in default mid; do
  if [ $v = mid ]; then export SMC_TWO_LEVEL_MID=1; fi
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/t_$v -o trace -- $B > $O/t_$v.log 2>&1
  python $R/tools/summarise_prof.py $O/t_$v | grep -v copyBuffer | head -8
  grep -h '^{' $O/t_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
  rm -rf $O/t_$v
done
