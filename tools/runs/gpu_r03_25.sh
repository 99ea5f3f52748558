#!/bin/bash
# round 3, call 25: large string batches interned and answered slice by slice (ACL_STRING_SLICES), intern pool workers that poll before they sleep
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout -s KILL 400 python -m pytest tests/test_engine_gpu.py tests/test_fullscale_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "string or named or entry_point" > $O/r03_25_tests.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -4 $O/r03_25_tests.log
[ $rc -ne 0 ] && exit 0
run() {
  timeout -s KILL 300 python bench.py --workload C4 --no-cpu --configs off --steps 20 2>$O/r03_25_err.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
sp=d['string_path']
print('slices=$ACL_STRING_SLICES value', round(d['value']/1e6,1), 'M/s | strings:', json.dumps({k:(round(v['decisions_per_s']/1e6,1) if isinstance(v,dict) and 'decisions_per_s' in v else v) for k,v in sp.items() if not isinstance(v,str)})[:600])"
}
nproc
for s in 0 1 0 1; do ACL_STRING_SLICES=$s run; done
