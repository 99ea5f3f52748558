#!/bin/bash
# round 3, call 15: one synchronisation per host-id call + string path: whole GPU suite, then the bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $O/r03_15_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r03_15_tests.log
timeout 400 python bench.py --steps 20 > $O/r03_15_bench.json 2> $O/r03_15_bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03_15_bench.json').read().strip().splitlines()[-1])
print('C4 value M/s', round(d['value']/1e6,1), 'device', round(d['device_resident']['decisions_per_s']/1e6,1), 'p50 ms', d['p50_batch_ms'], 'roofline', round(d['roofline']['frac'],3), 'parity', d['parity'])
for m,row in d['string_path']['sizes'].items():
    print('C4 strings', m, {k:(round(v['decisions_per_s']/1e6,1), round(v['ms_per_batch'],4), v['answers_equal_id_path']) for k,v in row.items()})
c2=d['configs']['C2']; print('C2 value M/s', round(c2['value']/1e6,1), 'device', round(c2['device_resident']['decisions_per_s']/1e6,1), 'lat', c2['latency']['p50_batch_ms'], 'parity', c2['parity'])
for m,row in c2['string_path']['sizes'].items():
    print('C2 strings', m, {k:(round(v['decisions_per_s']/1e6,1), round(v['ms_per_batch'],4), v['answers_equal_id_path']) for k,v in row.items()})
c3=d['configs']['C3']; print('C3', round(c3['value']), c3['p50_single_lookup_ms'], c3['parity'])
print('single', d['single_checks'])
P
