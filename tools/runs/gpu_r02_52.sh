#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -3
timeout 300 python bench.py --no-cpu --configs off 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value M/s', round(d['value']/1e6,1), 'device', round(d['device_resident']['decisions_per_s']/1e6,1), 'levels', d['levels'], 'string', round(d['string_path']['decisions_per_s']/1e6,1))"
