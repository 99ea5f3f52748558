#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 300 python bench.py --sharded on --logical-shards 8 --no-cpu --configs off --steps 5 --legs device > $O/r02_34_sharded8.json 2> $O/r02_34_sharded8.err; echo "sharded leg rc=$?"
python - <<PY
import json
d=json.loads(open("$O/r02_34_sharded8.json").read().strip().splitlines()[-1])
s=d.get("sharded",{})
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ("decisions_per_s","ms_per_batch","levels","mismatches_vs_replica","error","exchanges_per_batch","host_syncs_per_batch")}) for k,v in s.items() if k in ("allgather","alltoall","native","error","mismatches_vs_replica","shards")})
PY
tail -3 $O/r02_34_sharded8.err
timeout 100 python -m pytest tests/test_callers_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -2
timeout 200 tools/bin/batcher_bench 1000 64 256 1024 2>&1 | tail -3
