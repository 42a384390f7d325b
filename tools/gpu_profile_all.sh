#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 --kernel-trace --stats + FETCH_SIZE / WRITE_SIZE passes
# (separate passes, as MI355X_MICROARCH.md prescribes) of every workload of the bench line, each on the
# bench.py command the summary names.  $1 = tag, $2 = legs (default: all).
#   profiles/<tag>_<leg>_summary.txt   the table tools/summarise_prof.py prints
#   profiles/traffic_<leg>.json        bench.py's roofline.traffic source (config + command inside)
# are what to copy out of gpurun_out/<tag>/.
TAG=${1:-r12p}
LEGS=${2:-"c2 c3_systematic c3_stratified c3_multinomial c4 c4_dense c4_collapsed c5 sqmc c2_strict c3_systematic_strict"}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for leg in $LEGS; do
  STEPS=100; PASSES="FETCH_SIZE;WRITE_SIZE"
  case $leg in
    c2)             ARGS="--workload c2"; STEPS=400; CFG='{"workload":"c2","log2N":20,"islands":1,"scheme":"systematic"}'
                    PASSES="FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY;SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" ;;
    c3_systematic)  ARGS="--workload c3 --scheme systematic";  CFG='{"workload":"c3","log2N":22,"islands":1,"scheme":"systematic"}' ;;
    c3_stratified)  ARGS="--workload c3 --scheme stratified";  CFG='{"workload":"c3","log2N":22,"islands":1,"scheme":"stratified"}' ;;
    c3_multinomial) ARGS="--workload c3 --scheme multinomial"; CFG='{"workload":"c3","log2N":22,"islands":1,"scheme":"multinomial"}' ;;
    c4)             ARGS="--workload c4"; STEPS=40; CFG='{"workload":"c4","log2N":20,"islands":1,"scheme":"systematic","collapsed":false}'
                    PASSES="FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY;SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE" ;;
    c4_dense)       ARGS="--workload c4 --dense"; STEPS=40; CFG='{"workload":"c4","log2N":20,"islands":1,"scheme":"systematic","collapsed":false,"dense":true}'
                    PASSES="FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY;SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE" ;;
    c4_collapsed)   ARGS="--workload c4 --collapsed"; STEPS=40; CFG='{"workload":"c4","log2N":20,"islands":1,"scheme":"systematic","collapsed":true}'
                    PASSES="FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY;SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE" ;;
    c5)             ARGS="--workload c5"; CFG='{"workload":"c5","log2N":18,"islands":32,"scheme":"systematic"}' ;;
    sqmc)           ARGS="--workload c2 --qmc"; STEPS=40; CFG='{"workload":"c2","log2N":20,"islands":1,"scheme":"systematic","qmc":true}' ;;
    c2_strict)      ARGS="--workload c2 --strict"; STEPS=200; CFG='{"workload":"c2","log2N":20,"islands":1,"scheme":"systematic","strict":true}' ;;
    c3_systematic_strict) ARGS="--workload c3 --scheme systematic --strict"; CFG='{"workload":"c3","log2N":22,"islands":1,"scheme":"systematic","strict":true}' ;;
    *) echo "unknown leg $leg"; continue ;;
  esac
  P=$O/prof_$leg
  mkdir -p $P
  CMD="python bench.py $ARGS --steps $STEPS --warmup 20 --reps 3 --no-cpu-baseline --no-other-workloads --no-profile"
  BENCH="python $R/bench.py $ARGS --steps $STEPS --warmup 20 --reps 3 --no-cpu-baseline --no-other-workloads --no-profile"
  timeout 150 rocprofv3 --kernel-trace --stats -d $P/trace -o trace -- $BENCH > $P/trace.log 2>&1
  IFS=';' read -ra PASSLIST <<< "$PASSES"
  for pass in "${PASSLIST[@]}"; do
    name=$(echo $pass | cut -d' ' -f1)
    timeout 150 rocprofv3 --pmc $pass --kernel-trace -d $P/pmc_$name -o pmc -- $BENCH > $P/pmc_$name.log 2>&1
  done
  python $R/tools/summarise_prof.py $P --config "$CFG" --summary "${TAG}_${leg}_summary.txt" \
      --command "rocprofv3 --kernel-trace --stats -- $CMD ; rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-trace -- $CMD" \
      > $O/${TAG}_${leg}_summary.txt 2>&1
  cp $P/traffic.json $O/traffic_$leg.json 2>/dev/null
  grep -h '^{' $P/trace.log | tail -1 > $O/${TAG}_${leg}_bench_under_rocprof.json
  echo "=== $leg"; grep -v "^== traffic" $O/${TAG}_${leg}_summary.txt | grep -v "k_f_collect\|k_flush\|k_normal_rvs\|k_fill\|k_init" | head -40
  rm -rf $P
done
