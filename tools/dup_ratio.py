#!/usr/bin/env python3
"""How many frontier entries of the forward walk are DUPLICATES (same request, same group state reached through
several parents)?  The kernels keep no per-request visited set on the forward path (cycles must end at depth 50 exactly
as SpiceDB's dispatch does), so a state reached twice is expanded twice.  Pure numpy over the C4 generator: per level,
entries with multiplicity (what the GPU walks) vs distinct (request, group) pairs (what a per-request dedup would walk).
usage: tools/dup_ratio.py [scale] [sample]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")]
from aclgpu import workloads  # noqa: E402


def csr(res, subj, n):
    o = np.argsort(res, kind="stable")
    ptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(ptr, res.astype(np.int64) + 1, 1)
    return np.cumsum(ptr), subj[o]


def expand(ptr, col, req, node):
    deg = ptr[node + 1] - ptr[node]
    r = np.repeat(req, deg)
    start = np.repeat(ptr[node], deg)
    off = np.arange(deg.sum()) - np.repeat(np.cumsum(deg) - deg, deg)
    return r, col[start + off]


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    sample = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    w = workloads.c4(scale=scale)
    E = {(e[0], e[1], e[2], e[3]): (e[4], e[5]) for e in w.edges}
    ngrp, npod, nns = w.nobjects["group"], w.nobjects["pod"], w.nobjects["namespace"]
    gg = csr(*E[("group", "member", "group", "member")], ngrp)
    pvg = csr(*E[("pod", "viewer", "group", "member")], npod)
    nvg = csr(*E[("namespace", "viewer", "group", "member")], nns)
    pod_ns = E[("pod", "namespace", "namespace", "")][1]
    res = w.res[:sample].astype(np.int64)
    req = np.arange(res.size)
    r1, g1 = expand(*pvg, req, res)
    r2, g2 = expand(*nvg, req, pod_ns[res].astype(np.int64))
    req, grp = np.concatenate([r1, r2]), np.concatenate([g1, g2]).astype(np.int64)
    tot_m = tot_d = 0
    print("level  entries(with multiplicity)  distinct (req, group)  ratio")
    for lvl in range(1, 8):
        if not req.size:
            break
        d = np.unique(req.astype(np.int64) << 32 | grp).size
        print(f"{lvl:5d}  {req.size:12d}  {d:12d}  {req.size / d:.4f}")
        tot_m += req.size
        tot_d += d
        req, grp = expand(*gg, req, grp)
        grp = grp.astype(np.int64)
    print(f"total  {tot_m}  {tot_d}  duplicate ratio {tot_m / tot_d:.4f} (entries per distinct state)")


if __name__ == "__main__":
    main()
