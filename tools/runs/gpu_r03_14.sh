#!/bin/bash
# round 3, call 14: string entry points (interning straight into pinned staging, 8-byte hash, name pointer in the slot, {pointer, length} items): parity, then bench with named objects
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 500 python -m pytest tests/test_engine_gpu.py tests/test_callers_gpu.py tests/test_list_filter.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -4
timeout 300 python bench.py --no-cpu --steps 20 --configs off > $O/r03_14_bench.json 2> $O/r03_14_bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03_14_bench.json').read().strip().splitlines()[-1])
print('value M/s', round(d['value']/1e6,1), 'setup', d['setup_s'])
for m,row in d['string_path']['sizes'].items():
    print(m, {k:(round(v['decisions_per_s']/1e6,1), round(v['ms_per_batch'],4), v['answers_equal_id_path']) for k,v in row.items()})
P
