#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + PMC passes of
# the headline bench.  Outputs under gpurun_out/prof_<tag>/; copy the summaries
# worth keeping into profiles/.
TAG=${1:-r01}
STEPS=${2:-400}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps $STEPS --warmup 50 --reps 3 --no-cpu-baseline --no-profile $EXTRA"
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
DEFAULT_PASSES="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY;FETCH_SIZE;WRITE_SIZE;SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE;TCC_HIT_sum TCC_MISS_sum"
IFS=';' read -ra PASSLIST <<< "${PASSES:-$DEFAULT_PASSES}"
for pass in "${PASSLIST[@]}"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $pass --kernel-trace -d $OUT/pmc_$name -o pmc -- $BENCH > $OUT/pmc_$name.log 2>&1
done
rocprofv3 -L > $OUT/counters_list.txt 2>&1
python $R/tools/summarise_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# the sqlite outputs are tens of MB: only the summaries travel back
rm -rf $OUT/trace $OUT/pmc_*/ 2>/dev/null; find $OUT -name "*.db" -delete 2>/dev/null
