#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 400 python bench.py --workload C5 --steps 6 --no-cpu 2>gpurun_out/r02_55_c5.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:(round(v,1) if isinstance(v,float) else v) for k,v in d.items() if k in ('metric','value','unit','ms_per_step','n_gpus')}, d.get('parity'), str(d.get('config'))[:200])"
tail -3 gpurun_out/r02_55_c5.err
timeout 200 python bench.py --workload C1 --steps 20 --no-cpu --configs off 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C1 value', round(d['value']/1e6,2), 'M/s', d['device_resident']['dominant_kernel'])"
