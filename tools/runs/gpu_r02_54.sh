#!/bin/bash
# final measurements of round 2: full GPU test suite, default bench line, rocprof summaries (C4, C5-size replica), write latency, batcher
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/r02_54_tests.log 2>&1; echo "tests rc=$?"
tail -4 $O/r02_54_tests.log
timeout 500 python bench.py > $O/r02_54_bench.json 2> $O/r02_54_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/r02_54_bench.json").read().strip().splitlines()[-1])
print("value M/s", round(d["value"]/1e6,1), d["host_ids"]["mode"], "| device", round(d["device_resident"]["decisions_per_s"]/1e6,1), d["device_resident"]["level_loop_launches_per_batch"], "| p50 ms", d.get("p50_batch_ms"), "| roofline", d["roofline"]["kernel"], round(d["roofline"]["kernel_avg_us"],1), round(d["roofline"]["frac"],3), "| parity", d.get("parity"))
print("string", round(d["string_path"]["decisions_per_s"]/1e6,1), "cpu", round(d["cpu_baseline"]["value"]), d["cpu_baseline"]["cores"])
for k,v in d.get("configs",{}).items(): print(k, {kk: (round(vv,1) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("value","unit","p50_batch_ms")}, v.get("roofline",{}).get("frac"), v.get("parity"))
PY
timeout 300 bash tools/prof_c4.sh r02_c4_v16 > /dev/null 2>&1
python tools/rocprof_summary.py r02_c4_v16 $O/prof/r02_c4_v16/stats/r_results.db $O/prof/r02_c4_v16/fetch/r_results.db $O/prof/r02_c4_v16/write/r_results.db --kernel k_check_local --out $O/profiles_r02 | grep -E "k_check_local|FETCH|WRITE|traffic"
timeout 400 python bench.py --workload C5 --replica --steps 10 --configs off > $O/r02_54_c5r_bench.json 2> $O/r02_54_c5r.err; echo "c5 replica bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/r02_54_c5r_bench.json").read().strip().splitlines()[-1])
print("C5R value M/s", round(d["value"]/1e6,1), "| device", round(d["device_resident"]["decisions_per_s"]/1e6,1), d["device_resident"]["dominant_kernel"], d["device_resident"]["level_loop_launches_per_batch"], "| roofline", round(d["roofline"]["kernel_avg_us"],1), round(d["roofline"]["frac"],3), "| parity", d.get("parity"), "snapshot", d["snapshot_bytes"])
PY
timeout 400 bash tools/prof_c4.sh r02_c5r_v5 --workload C5 --replica > /dev/null 2>&1
python tools/rocprof_summary.py r02_c5r_replica_v5 $O/prof/r02_c5r_v5/stats/r_results.db $O/prof/r02_c5r_v5/fetch/r_results.db $O/prof/r02_c5r_v5/write/r_results.db --workload C5R --kernel k_check_local --out $O/profiles_r02 | grep -E "k_check_local|FETCH|WRITE|traffic"
timeout 400 python tools/write_latency.py 2>&1 | tail -1 > $O/r02_54_write_latency.json; cut -c1-400 $O/r02_54_write_latency.json
timeout 200 tools/bin/batcher_bench 1000 64 256 1024 2>&1 | tee $O/r02_54_batcher.txt | tail -4
