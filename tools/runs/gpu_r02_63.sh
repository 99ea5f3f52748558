#!/bin/bash
# every concurrent call shape at once against one engine (tools/engine_stress.cpp): answers compared; then the same under ThreadSanitizer
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 60 tools/bin/engine_stress 4 2>&1 | tail -3 | tee $O/r02_63_engine_stress.txt
export TSAN_OPTIONS="halt_on_error=0 exitcode=0"
timeout 100 tools/bin/tsan/engine_stress 3 > $O/r02_63_tsan.out 2> $O/r02_63_tsan.err; echo "tsan rc=$?" | tee -a $O/r02_63_engine_stress.txt
tail -2 $O/r02_63_tsan.out | tee -a $O/r02_63_engine_stress.txt
echo "ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' $O/r02_63_tsan.err)" | tee -a $O/r02_63_engine_stress.txt
grep -h "SUMMARY" $O/r02_63_tsan.err | sort | uniq -c | sort -rn | head -25 | tee -a $O/r02_63_engine_stress.txt
head -c 6000 $O/r02_63_tsan.err > $O/r02_63_tsan_head.err
