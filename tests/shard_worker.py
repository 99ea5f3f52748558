"""One rank of the world_size-N gloo protocol test (spawned by tests/test_sharded_gloo.py).

usage: python shard_worker.py <cases.json> <out.json>     (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in env)
Runs aclgpu.sharded.ShardedEngine (the product's SPMD protocol) over TorchComm(gloo) with the CpuShard test
double as the shard; rank 0 writes every rank-0 result, every rank asserts its results equal rank 0's."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from aclgpu.sharded import ShardedEngine, TorchComm  # noqa: E402
from tests.shard_double import CpuShard, Universe  # noqa: E402


def main():
    cases = json.load(open(sys.argv[1]))
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = TorchComm(device="cpu")
    results = []
    for case in cases:
        u = Universe(case["schema"], [tuple(t) for t in case["tuples"]])
        items = u.items([tuple(q) for q in case["queries"]])
        for (_rt, _pm, st, sid, _sr) in case["lookups"]:
            u.ensure(st, sid)
        sh = CpuShard(u, rank, world)
        eng = ShardedEngine(sh, comm, export_entries=case.get("export_entries", 1 << 12), exchange=os.environ.get("ACL_EXCHANGE", "allgather"))
        perm, err = eng.check_bulk_ids(items)
        res = {"perm": perm.tolist(), "err": err.tolist(), "levels": eng.levels_last, "exchanges": eng.exchanges, "lookups": [],
               "owners": {t: sh.owner_of_type(t) for t in u.types}, "cap": eng.cap}
        for (rt, pm, st, sid, sr) in case["lookups"]:
            bm = eng.lookup_ids_batch(rt, pm, st, sr, [u.oid[st][sid]])
            bits = np.unpackbits(bm[0].numpy().view(np.uint8), bitorder="little")
            res["lookups"].append(sorted(u.names[rt][i] for i in np.flatnonzero(bits)))
        # every rank must hold the same answers
        blob = [None] * world
        dist.all_gather_object(blob, {k: res[k] for k in ("perm", "err", "lookups")})
        assert all(b == blob[0] for b in blob), "ranks disagree"
        results.append(res)
    if rank == 0:
        json.dump(results, open(sys.argv[2], "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
