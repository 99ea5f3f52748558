#!/bin/bash
# round 4, call 1: FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/counter_calib.hip), then the round's baseline line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/calib
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o r -- $R/tools/bin/counter_calib > $O/run_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o r -- $R/tools/bin/counter_calib > $O/run_write.log 2>&1
ls -R $O | head -30
python $R/tools/calib_summary.py $O/run_fetch.log $(ls $O/fetch/*.db | head -1) $(ls $O/write/*.db | head -1) --out $O/r04_counter_calibration
cd $R
timeout 400 python bench.py > gpurun_out/r04_baseline_bench.log 2>&1; tail -1 gpurun_out/r04_baseline_bench.log | cut -c1-1500
