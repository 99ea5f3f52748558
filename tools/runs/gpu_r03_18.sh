#!/bin/bash
# round 3, call 18: C5 mixed stream on 8 logical shards (emulated on one GPU): level loops inside the library vs the host-driven protocol; sharded leg of the C4 bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'M decisions/s', round(d['value']/1e6,2), 'ms/check batch', round(d['ms_per_check_batch'],3), 'ms/filter', round(d['ms_per_filter_request'],3), 'levels', d['levels'], 'parity', d['parity'])"; }
timeout 400 python bench.py --workload C5 --scale 0.3 --logical-shards 8 --steps 12 2>$O/r03_18_c5.err | show "C5 x0.3 native all-to-all"
timeout 400 python bench.py --workload C5 --scale 0.3 --logical-shards 8 --steps 12 --native-loop off 2>>$O/r03_18_c5.err | show "C5 x0.3 host-driven all-gather"
ACL_SHARD_A2A=0 timeout 400 python bench.py --workload C5 --scale 0.3 --logical-shards 8 --steps 12 2>>$O/r03_18_c5.err | show "C5 x0.3 native all-gather"
timeout 600 python bench.py --workload C5 --scale 1.0 --logical-shards 8 --steps 10 > $O/r03_18_c5_full.json 2>>$O/r03_18_c5.err; cat $O/r03_18_c5_full.json | show "C5 x1.0 native all-to-all"
timeout 400 python bench.py --no-cpu --steps 10 --configs off --sharded on --logical-shards 8 > $O/r03_18_c4_sharded.json 2>>$O/r03_18_c5.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03_18_c4_sharded.json').read().strip().splitlines()[-1])
sh=d['sharded']
for m in ('allgather','alltoall','native'):
    if m in sh: print('C4 sharded leg', m, {k:(round(v,2) if isinstance(v,float) else v) for k,v in sh[m].items() if k in ('decisions_per_s','ms_per_batch','levels','exchanges_per_batch','mismatches_vs_replica','error')})
P
tail -3 $O/r03_18_c5.err | grep -v amdgpu
