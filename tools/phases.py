#!/usr/bin/env python3
"""usage (GPU box): ACLGPU_LIB=.../libaclgpu_phases.so python tools/phases.py [C4|C2] -- where the single-launch walk's wave-time goes.
Needs the variant built with -DACL_PROFILE_PHASES=1 (tools/build_variant.sh phases -DACL_PROFILE_PHASES=1)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
import torch  # noqa: E402
import aclgpu  # noqa: E402
from aclgpu import workloads  # noqa: E402

NAMES = ["other", "entries wait", "parent gathers wait", "task creation", "flush prologue (scan, head bits)", "edges wait", "buckets wait + hash",
         "compare + push", "generic interpreter (whole segments)", "level barriers", "seeding"]
w = getattr(workloads, (sys.argv[1] if len(sys.argv) > 1 else "C4").lower())()
e = aclgpu.Engine(w.schema)
w.load(e)
rt, perm, st = w.check
items = e.make_items(rt, perm, w.res, st, "", w.subj)
n = len(items)
d_items = torch.from_numpy(items.view(np.uint8).copy()).cuda()
d_perm = torch.zeros(n, dtype=torch.uint8, device="cuda")
d_err = torch.zeros(n, dtype=torch.int32, device="cuda")
lib = C.CDLL(os.environ["ACLGPU_LIB"])
out = (C.c_ulonglong * 16)()
for _ in range(3):
    e.check_bulk_ids_device(d_items.data_ptr(), n, d_perm.data_ptr(), d_err.data_ptr())
torch.cuda.synchronize()
lib.acl_debug_phase_cycles(out)
K = 5
e.set_timing(True)
e.stats_reset()
for _ in range(K):
    e.check_bulk_ids_device(d_items.data_ptr(), n, d_perm.data_ptr(), d_err.data_ptr())
torch.cuda.synchronize()
st_ = e.stats()
lib.acl_debug_phase_cycles(out)
tot = sum(out[i] for i in range(len(NAMES)))
print(f"{w.name}: {n} items, kernel {1e3 * st_['local_ms'] / K:.1f} us per batch (instrumented), {tot / K / 1e6:.1f} M wave-cycles per batch")
for i, nm in enumerate(NAMES):
    print(f"  {nm:40s} {100.0 * out[i] / max(tot, 1):5.1f} %")
# counters of the deep levels' fast path (slots 12-15): entries read, entries of requests answered meanwhile, segments, pairs that would fit one segment
if out[12]:
    print(f"  deep-level entries read per batch {out[12] / K / 1e6:.2f} M, of which dead on arrival {100.0 * out[13] / out[12]:.1f} %; segments {out[14] / K / 1e3:.0f} k, "
          f"pairs whose live entries fit ONE segment {100.0 * 2 * out[15] / max(out[14], 1):.1f} % of the segments")
e.close()
