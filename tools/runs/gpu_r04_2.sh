#!/bin/bash
# round 4, call 2: the GPU suite with the combine kernels (tests/test_combine_gpu.py first), the read-request size split of the calibration
# kernels (is TCC_BUBBLE alive on gfx950?), the default bench line with the monotone kernels after the CMB templating
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_combine_gpu.py -q -x --timeout 600 2>&1 | tail -25
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 --deselect tests/test_combine_gpu.py 2>&1 | tail -15
O=$R/gpurun_out/calib2; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --pmc TCC_BUBBLE_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/rd -o r -- $R/tools/bin/counter_calib > $O/run.log 2>&1; tail -3 $O/run.log )
python - <<PY
import sqlite3,glob,collections
db=glob.glob("$O/rd/*.db")
if db:
    con=sqlite3.connect(db[0])
    rows=con.execute("select kernel_name,counter_name,value from counters_collection order by start").fetchall()
    acc=collections.OrderedDict()
    for k,c,v in rows:
        if not any(x in k for x in ("k_gather","k_stream","k_scatter","k_append")): continue
        acc.setdefault(k.split("(")[0][-40:],{}).setdefault(c,[]).append(v)
    for k,d in acc.items(): print(k, {c:(v[0],v[1] if len(v)>1 else None) for c,v in d.items()})
PY
timeout 400 python bench.py > gpurun_out/r04_bench_2.log 2>&1; tail -1 gpurun_out/r04_bench_2.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('value %.1f M/s kernel %.1f us frac %.3f C2 %.2f G/s C3 %.2f M lookups/s' % (d['value']/1e6, d['roofline']['kernel_avg_us'], d['roofline']['frac'], d['configs']['C2']['value']/1e9, d['configs']['C3']['value']/1e6))"
