mkdir -p gpurun_out/r6d
python -m pytest -m gpu -x -q tests/test_lookup_local_gpu.py tests/test_engine_gpu.py tests/test_combine_gpu.py tests/test_callers_gpu.py tests/test_write_path_gpu.py "tests/test_fullscale_gpu.py::test_c3_full_all_power_users" > gpurun_out/r6d/lookup_tests.txt 2>&1; tail -5 gpurun_out/r6d/lookup_tests.txt
ACL_DEBUG_REV=1 python tools/lookup_big_probe.py 1.0 2>&1 | grep -v "^\[aclgpu\] lookup [1-9]" | awk '!seen[$0]++' | tee gpurun_out/r6d/lookup_big_probe_after.txt | grep -v "deferred rows" 
grep "deferred rows" gpurun_out/r6d/lookup_big_probe_after.txt | tail -3
python tools/lookup_latency.py 2>&1 | tail -12 | tee gpurun_out/r6d/lookup_latency_c3.txt
