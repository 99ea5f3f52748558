mkdir -p gpurun_out/r6d
python tools/lookup_big_probe.py 1.0 2>&1 | tee gpurun_out/r6d/lookup_big_probe_1m.txt
ACL_REV_BIG_ROWS=0 python tools/lookup_big_probe.py 1.0 2>&1 | tee gpurun_out/r6d/lookup_big_probe_1m_oneblock.txt
echo "== C3, rows in LDS (default)"; python tools/lookup_latency.py 2>&1 | tail -4 | tee gpurun_out/r6d/lat_c3_lds.txt
echo "== C3, rows as bytes in HBM + deferral + chip-wide fold (ACL_REV_LDS_ROWS=0)"; ACL_REV_LDS_ROWS=0 python tools/lookup_latency.py 2>&1 | tail -4 | tee gpurun_out/r6d/lat_c3_big.txt
echo "== C3, ACL_REV_LDS_ROWS=0 ACL_REV_DEFER_MIN=1000000 (bytes + chip-wide fold, nothing deferred)"; ACL_REV_LDS_ROWS=0 ACL_REV_DEFER_MIN=1000000 python tools/lookup_latency.py 2>&1 | tail -4 | tee gpurun_out/r6d/lat_c3_big_nodefer.txt
