#!/bin/bash
# usage (GPU box, via gpurun): tools/profile_round.sh <round tag, e.g. r06> -- everything the round's profiles/ entries are made of, in one call:
#   per configuration (C4, C5R = the 100 M-relationship replica, C2, C3): rocprofv3 --kernel-trace --stats of bench.py's device leg and separate --pmc FETCH_SIZE /
#   --pmc WRITE_SIZE passes (tools/prof_c4.sh), summarised by tools/rocprof_summary.py into gpurun_out/profiles_<tag>/<tag>_<cfg>.md (+ traffic.json);
#   the issue / cache counter sets of the dominant kernel on C4 and C5R (one --pmc pass per set, --kernel-trace only: tools/pmc.sh), summarised into <tag>_pmc_<cfg>.md.
# Copy what should be judged from gpurun_out/profiles_<tag>/ into profiles/ (tracked).
RT=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$RT
mkdir -p $OUT
db() { find $R/gpurun_out/prof/$1 -name "*.db" | head -1; }
for CFG in C4 C5R C2 C3; do
  ARGS="--workload $CFG"; KERN=k_check_local; ITEMS=262144
  [ $CFG = C5R ] && ARGS="--workload C5 --replica"
  [ $CFG = C2 ] && ITEMS=65536
  [ $CFG = C3 ] && { KERN=k_rev_local; ITEMS=0; }
  T=${RT}_$(echo $CFG | tr A-Z a-z)
  bash $R/tools/prof_c4.sh $T $ARGS > /dev/null 2>&1
  python $R/tools/rocprof_summary.py $T "$(db $T/stats)" "$(db $T/fetch)" "$(db $T/write)" --workload $CFG --steps 5 --kernel $KERN --items $ITEMS --out $OUT > /dev/null 2>&1
  head -12 $OUT/$T.md
done
# the 90 / 10 Check + Filter stream on the replica: the three launches of a LookupResources over the 8.45 M-pod type (kernel trace only)
T=${RT}_c5r_stream
mkdir -p $R/gpurun_out/prof/$T
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/$T/stats -o r -- python $R/bench.py --workload C5 --replica --stream --steps 5 --warmup 2 --no-cpu --legs device --configs off > $R/gpurun_out/prof/$T/stats.log 2>&1)
python $R/tools/rocprof_summary.py $T "$(db $T/stats)" --workload C5R-stream --steps 5 --kernel k_rev_local --out $OUT > /dev/null 2>&1
head -14 $OUT/$T.md
for CFG in C4 C5R; do
  ARGS=""; [ $CFG = C5R ] && ARGS="--workload C5 --replica"
  TAG=${RT}_pmc_$CFG
  i=0
  for SET in "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    PMC_TIMEOUT=150 bash $R/tools/pmc.sh $TAG/s$i "$SET" $ARGS > /dev/null 2>&1
  done
  python $R/tools/pmc_r05_summary.py $R/gpurun_out/prof/$TAG $CFG "round-6" "tools/profile_round.sh" > $OUT/${RT}_pmc_$(echo $CFG | tr A-Z a-z).md
  sed -n '/## derived/,$p' $OUT/${RT}_pmc_$(echo $CFG | tr A-Z a-z).md
done
rm -rf $R/gpurun_out/prof/${RT}_*  # (the raw databases: tens of MB, summarised above)
