#!/usr/bin/env python3
"""usage (GPU box): ACLGPU_LIB=<variant .so> python tools/ab_kernel.py [C4|C5R|C2] [launches] -- device-resident kernel time of ONE library variant.

Prints one line: variant, mean / min kernel us over `launches` launches (HIP events inside the engine), HAS count and a checksum of every answer --
tools/ab_variants.sh runs the variants found in lib/ round-robin and the checksums must agree between them (the baseline's answers are the
oracle-checked ones of tests/ and bench.py).  The generated workload is cached under /tmp so that the second variant does not pay for it again."""
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
import torch  # noqa: E402
import aclgpu  # noqa: E402
from aclgpu import _lib, workloads  # noqa: E402

name = (sys.argv[1] if len(sys.argv) > 1 else "C4").upper()
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cache = f"/tmp/ab_{name}.npz"
t0 = time.time()
if name == "C5R":
    gen = lambda: workloads.c5(scale=1.0)  # noqa: E731  the 100 M-relationship graph as ONE replica
elif name == "C2":
    gen = lambda: workloads.c2()  # noqa: E731
else:
    gen = lambda: workloads.c4()  # noqa: E731
if os.path.exists(cache):
    z = np.load(cache, allow_pickle=True)
    w = workloads.Workload(name, str(z["schema"]))
    w.edges = [(str(a), str(b), str(c), str(d), z[f"r{i}"], z[f"s{i}"]) for i, (a, b, c, d) in enumerate(z["heads"])]
    w.check, w.res, w.subj = tuple(str(x) for x in z["check"]), z["res"], z["subj"]
else:
    w = gen()
    np.savez(cache, schema=w.schema, heads=np.array([e[:4] for e in w.edges]), check=np.array(w.check), res=w.res, subj=w.subj,
             **{f"r{i}": e[4] for i, e in enumerate(w.edges)}, **{f"s{i}": e[5] for i, e in enumerate(w.edges)})
t_gen = time.time() - t0
e = aclgpu.Engine(w.schema)
w.load(e)
rt, perm, st = w.check
items = e.make_items(rt, perm, w.res, st, "", w.subj)
n = len(items)
d_items = torch.from_numpy(items.view(np.uint8).copy()).cuda()
d_perm = torch.zeros(n, dtype=torch.uint8, device="cuda")
d_err = torch.zeros(n, dtype=torch.int32, device="cuda")
for _ in range(3):
    e.check_bulk_ids_device(d_items.data_ptr(), n, d_perm.data_ptr(), d_err.data_ptr())
torch.cuda.synchronize()
e.set_timing(True)
per = []
for _ in range(K):
    e.stats_reset()
    e.check_bulk_ids_device(d_items.data_ptr(), n, d_perm.data_ptr(), d_err.data_ptr())
    torch.cuda.synchronize()
    per.append(1e3 * e.stats()["local_ms"])
st_ = e.stats()
p = d_perm.cpu().numpy()
er = d_err.cpu().numpy()
tag = os.path.basename(_lib.LIB_PATH).replace("libaclgpu", "").replace(".so", "").strip("_") or "main"
env = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("ACL_") and k != "ACLGPU_LIB")
print(f"{name} {tag:12s} {env:24s} kernel us mean {np.mean(per):7.1f} min {np.min(per):7.1f} median {np.median(per):7.1f} | HAS {int((p == 2).sum())} "
      f"crc {zlib.crc32(p.tobytes()) ^ zlib.crc32(er.tobytes()):08x} | local passes {st_['local_passes']} | gen {t_gen:.0f}s load {time.time() - t0 - t_gen:.0f}s", flush=True)
e.close()
