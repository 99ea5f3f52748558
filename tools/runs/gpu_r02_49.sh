#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for K in 5 10 40; do
timeout 300 python bench.py --gpus 1 --steps $K --warmup 2 --no-cpu --configs off 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps $K: value M/s', round(d['value']/1e6,1), 'ms_per_step', round(d['ms_per_step'],4), 'device', round(d['device_resident']['decisions_per_s']/1e6,1))"
done
timeout 100 python bench.py --pipeline submit --steps 20 --no-cpu --configs off 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('submit: value M/s', round(d['value']/1e6,1))"
