#!/bin/bash
# end of round 2: full GPU test suite + the default bench line (with the single_checks leg)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/r02_61_tests.log 2>&1; echo "tests rc=$?"
tail -2 $O/r02_61_tests.log
timeout 400 python bench.py > $O/r02_61_bench.json 2> $O/r02_61_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/r02_61_bench.json").read().strip().splitlines()[-1])
print("value M/s", round(d["value"]/1e6,1), "| device", round(d["device_resident"]["decisions_per_s"]/1e6,1), "| p50 ms", d.get("p50_batch_ms"), "| roofline", d["roofline"]["kernel"], round(d["roofline"]["kernel_avg_us"],1), round(d["roofline"]["frac"],3), "| parity", d.get("parity"))
print("string", round(d["string_path"]["decisions_per_s"]/1e6,1), "cpu", round(d["cpu_baseline"]["value"]), d["cpu_baseline"]["cores"])
for k,v in d.get("configs",{}).items(): print(k, {kk: (round(vv,1) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("value","unit","p50_batch_ms")}, v.get("roofline",{}).get("frac"), v.get("parity"))
print("single_checks", json.dumps(d.get("single_checks")))
PY
