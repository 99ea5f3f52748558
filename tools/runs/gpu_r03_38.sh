#!/bin/bash
# round 3, call 38: the default bench with roofline.traffic measured in the run (two rocprofv3 --pmc passes of its own device leg)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
( time timeout -s KILL 900 python bench.py > $O/r03_38_bench.json 2> $O/r03_38_bench.err ) 2>&1 | grep real
python - <<P
import json
d=json.loads(open('$O/r03_38_bench.json').read().strip().splitlines()[-1])
print('value %.1f M/s kernel %.1f us frac %.3f traffic %s' % (d['value']/1e6, d['roofline']['kernel_avg_us'], d['roofline']['frac'], d['roofline']['traffic']))
print(json.dumps(d['roofline']['traffic_detail'], indent=1))
P
tail -3 $O/r03_38_bench.err | grep -v amdgpu.ids
