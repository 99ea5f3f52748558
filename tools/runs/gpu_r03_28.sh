#!/bin/bash
# round 3, call 28: name-table slot layout A/B (old: {tag,id,pointer}; new: cache-line slots with the name inline), intern threads 16 / 32; 40 calls per point
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
run() {
  timeout -s KILL 300 python bench.py --workload C4 --no-cpu --configs off --steps 10 2>$O/r03_28_err.txt | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
sp=d['string_path']['sizes']
print('$1 threads=${ACL_INTERN_THREADS:-16} | ' + ' | '.join(k+' views best %.1f p50 %.1f  c_str best %.1f p50 %.1f M/s' % (v['views']['decisions_per_s']/1e6, int(k)/v['views']['p50_ms']/1e3, v['c_strings']['decisions_per_s']/1e6, int(k)/v['c_strings']['p50_ms']/1e3) for k,v in sp.items()), '| eq', all(v['views']['answers_equal_id_path'] and v['c_strings']['answers_equal_id_path'] for v in sp.values()))"
}
export ACL_STRING_SLICES=0
for i in 1 2 3; do
  ACLGPU_LIB=$R/spicedb-kubeapi-proxy_amd/lib/libaclgpu_oldslots.so run old
  run new
  ACLGPU_LIB=$R/spicedb-kubeapi-proxy_amd/lib/libaclgpu_oldslots.so ACL_INTERN_THREADS=32 run old
  ACL_INTERN_THREADS=32 run new
done
