#!/bin/bash
# the sharded leg isolated in a child process (bench.py --sharded-isolate): plumbing check on one GPU with logical shards
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 280 python bench.py --scale 0.2 --steps 5 --no-cpu --configs off --sharded on --logical-shards 4 --sharded-isolate on > $O/r02_62_iso.json 2> $O/r02_62_iso.err; echo "rc=$?"
wc -l $O/r02_62_iso.json
python - <<PY
import json
d=json.loads(open("$O/r02_62_iso.json").read().strip().splitlines()[-1])
s=d.get("sharded",{})
print("value", round(d["value"]/1e6,1), "sharded keys", sorted(s.keys()))
for m in ("allgather","alltoall","native"):
    if m in s: print(m, {k:(round(v,2) if isinstance(v,float) else v) for k,v in s[m].items() if k in ("decisions_per_s","ms_per_batch","levels","mismatches_vs_replica","error")})
print(s.get("error"), s.get("isolated"))
PY
tail -3 $O/r02_62_iso.err
