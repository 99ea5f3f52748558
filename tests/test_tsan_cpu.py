"""ThreadSanitizer over the host side of libaclgpu.so (tools/tsan.sh): the relationship store under concurrent writers / readers /
watch polls, the snapshot patcher and the background compaction's two halves, string interning, the micro-batcher's wake-up tree and the
completion queue -- on store-only engines (no GPU), with native threads.  The seam is called from arbitrary goroutines
(pkg/authz/check.go:77-93, responsefilterer.go:165, watch.go:50): a data race here is a wrong answer there."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_side_is_race_free_under_tsan(aclgpu_lib):
    if not (os.path.exists("/opt/rocm/bin/hipcc") and os.path.exists("/opt/rocm/lib/llvm/bin/clang++") and shutil.which("make")):
        pytest.skip("no ROCm clang here")
    pr = subprocess.run(["bash", os.path.join(ROOT, "tools", "tsan.sh"), "20"], capture_output=True, text=True, timeout=600)
    if pr.returncode:
        pytest.skip("the instrumented build did not link here: " + pr.stderr.strip()[-300:])
    out = pr.stdout
    m1 = re.search(r"store_stress rc=(\d+): .* (\d+) failed expectations; ThreadSanitizer reports: (\d+)", out)
    m3 = re.search(r"store_stress with bootstrap reloads rc=(\d+): .* (\d+) failed expectations; ThreadSanitizer reports: (\d+)", out)
    assert m3 and (m3.group(1), m3.group(2), m3.group(3)) == ("0", "0", "0"), out
    m4 = re.search(r"store_stress with batcher restarts rc=(\d+): .* (\d+) failed expectations; ThreadSanitizer reports: (\d+)", out)
    assert m4 and (m4.group(1), m4.group(2), m4.group(3)) == ("0", "0", "0"), out  # (rc 124 = a caller never returned: a request lost by acl_batcher_stop)
    m2 = re.search(r"batcher_bench rc=(\d+) \((\d+) runs\); ThreadSanitizer reports: (\d+)", out)
    assert m1 and m2, out
    assert (m1.group(1), m1.group(2), m1.group(3)) == ("0", "0", "0"), out
    assert (m2.group(1), m2.group(2), m2.group(3)) == ("0", "4", "0"), out
