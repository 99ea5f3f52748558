#!/bin/bash
# round 3, call 1: the single-launch reverse walk (k_rev_local) -- its own tests first, then the whole GPU suite, then C3 A/B against the level loop
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 300 python -m pytest tests/test_lookup_local_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/r03_2_lookup_tests.log 2>&1; echo "lookup tests rc=$?"
tail -15 $O/r03_2_lookup_tests.log
timeout 400 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/r03_2_tests.log 2>&1; echo "tests rc=$?"
tail -5 $O/r03_2_tests.log
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1 C3 lookups/s', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel ms', round(d['kernel_ms_per_step'],4), 'launches', d['launches_per_step'], 'single p50 ms', round(d['p50_single_lookup_ms'],4), 'pageable', d['pageable_result_buffers'], 'kernel', d['roofline']['kernel'], 'avg us', round(d['roofline']['kernel_avg_us'],1), 'levels', d['reverse_levels'])"; }
for i in 1 2; do timeout 120 python bench.py --workload C3 --no-cpu --steps 40 2>$O/r03_2_c3.err | tail -1 | show local; done
for i in 1 2; do ACL_REV_LOCAL=0 timeout 120 python bench.py --workload C3 --no-cpu --steps 40 2>>$O/r03_2_c3.err | tail -1 | show levelloop; done
timeout 200 python bench.py --workload C3 --steps 40 > $O/r03_2_c3_full.json 2>>$O/r03_2_c3.err; echo "c3 full rc=$?"; tail -c 1500 $O/r03_2_c3_full.json
