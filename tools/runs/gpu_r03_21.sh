#!/bin/bash
# round 3, call 21: the default bench line (what the driver runs) + test durations
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03_21_bench.json 2> $O/r03_21_bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03_21_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','n_gpus','ms_per_step','p50_batch_ms','setup_s')})
print('roofline', {k:v for k,v in d['roofline'].items() if k in ('achieved','frac','traffic','kernel','kernel_avg_us')}, 'parity', d['parity'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
print('device', d['device_resident']['decisions_per_s'], 'strings', d['string_path']['decisions_per_s'], d['string_path']['answers_equal_id_path'])
for k,c in d['configs'].items():
    if isinstance(c, dict): print(k, c.get('value'), c.get('roofline',{}).get('frac'), c.get('roofline',{}).get('traffic'), c.get('parity'), c.get('cpu_baseline',{}).get('value'))
    else: print(k, c)
P

timeout 400 python -m pytest tests/test_fullscale_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider --durations=5 2>&1 | grep -E "s call|passed|failed" | head
