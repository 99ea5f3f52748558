mkdir -p gpurun_out/r6d
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_lb -o lb -- python $GRAFT_REPO_ROOT/tools/lookup_big_probe.py 1.0 > /tmp/prof_lb.log 2>&1
DB=$(find /tmp/prof_lb -name "*.db" | head -1)
python - "$DB" > $GRAFT_REPO_ROOT/gpurun_out/r6d/lookup_big_rocprof.txt <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    short = name.replace("acl::(anonymous namespace)::", "").split("(")[0]
    print(f"{short:40s} calls {calls:6d} total us {total/1e3:10.1f} avg us {avg/1e3:9.2f} {pct:6.2f} %")
rows = con.execute("select name,start,duration,grid_x,workgroup_x from kernels where name like '%k_rev_%' order by start").fetchall()
print("last 18 reverse launches (the all-5 batch and the deep-2 singles before it):")
for name, start, dur, gx, wx in rows[-18:]:
    print(f"  {name.replace('acl::(anonymous namespace)::','').split('(')[0]:20s} grid {gx:8d} wg {wx:5d}  {dur/1e3:9.2f} us")
PY
cat $GRAFT_REPO_ROOT/gpurun_out/r6d/lookup_big_rocprof.txt
