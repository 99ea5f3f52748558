#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
export ACL_SKIP_C5_FULL=1
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $O/r02_20_tests.log 2>&1; echo "tests rc=$?"
tail -5 $O/r02_20_tests.log
timeout 400 python bench.py > $O/r02_20_bench.json 2> $O/r02_20_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/r02_20_bench.json").read().strip().splitlines()[-1])
print("value M/s", round(d["value"]/1e6,1), d["host_ids"]["mode"] if "host_ids" in d else "", "| device", round(d["device_resident"]["decisions_per_s"]/1e6,1), "| p50 ms", d.get("p50_batch_ms"), "| roofline", d["roofline"]["kernel"], d["roofline"]["kernel_avg_us"], d["roofline"]["frac"], "| parity", d.get("parity"))
print("string", round(d["string_path"]["decisions_per_s"]/1e6,1), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
for k,v in d.get("configs",{}).items(): print(k, {kk: (round(vv,1) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("value","unit","p50_batch_ms")}, v.get("roofline",{}).get("frac"), v.get("parity"))
PY
for CL in 1 2 3; do
  timeout 100 python bench.py --no-cpu --configs off --steps 40 --callers $CL 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('--callers $CL: host-id M/s', round(d['value']/1e6,1), 'single-call p50 ms', round(d['latency']['p50_batch_ms'],4), 'device M/s', round(d['device_resident']['decisions_per_s']/1e6,1))"
done 2>&1 | tee $O/r02_20_modes.txt
bash tools/prof_c4.sh r02_c4_v12 > /dev/null 2>&1
python tools/rocprof_summary.py r02_c4_v12 $O/prof/r02_c4_v12/stats/r_results.db $O/prof/r02_c4_v12/fetch/r_results.db $O/prof/r02_c4_v12/write/r_results.db --kernel k_check_local --out $O/profiles_r02 | tail -22
