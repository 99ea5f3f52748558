#!/usr/bin/env python3
"""print per-launch PMC values of the last N k_expand launches from a rocprofv3 rocpd db"""
import sqlite3, sys, collections
db, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 7
kern = sys.argv[3] if len(sys.argv) > 3 else "k_"
con = sqlite3.connect(db)
rows = con.execute(f"select dispatch_id, counter_name, value, duration from counters_collection where kernel_name like '%{kern}%' order by dispatch_id").fetchall()
by = collections.OrderedDict()
for d, c, v, dur in rows:
    by.setdefault(d, {"dur_us": dur / 1e3})[c] = v
keys = list(by.keys())[-n:]
names = sorted({c for k in keys for c in by[k] if c != "dur_us"})
print("| # | dur_us | " + " | ".join(names) + " |")
print("|---|---|" + "---|" * len(names))
for i, k in enumerate(keys):
    print(f"| {i+1} | {by[k]['dur_us']:.1f} | " + " | ".join(f"{by[k].get(c, float('nan')):.4g}" for c in names) + " |")
