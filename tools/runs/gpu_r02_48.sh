#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value M/s', round(d['value']/1e6,1), 'steps', d['steps'], 'ms_per_step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'cpu', round(d['cpu_baseline']['value']), list(d.keys()))"
