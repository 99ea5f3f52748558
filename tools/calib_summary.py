#!/usr/bin/env python3
"""Joins tools/counter_calib's launches (its `CALIB name table read write` lines, in launch order) with the FETCH_SIZE / WRITE_SIZE
values of two rocprofv3 --pmc passes (rocpd sqlite) and writes profiles/r04_counter_calibration.{md,json}.

usage: calib_summary.py <run.log> <fetch.db> <write.db> [--out profiles/r04_counter_calibration]

The factors it prints are `algorithmic bytes / (counter KiB x 1024)`: what a counter value has to be multiplied by to become bytes
in that access shape.  bench.py's measure_traffic and tools/rocprof_summary.py read the json (keys `fetch_factor`, `write_factor`)."""
import argparse
import json
import os
import sqlite3
import statistics


def rows(db, counter):
    con = sqlite3.connect(db)
    try:
        return con.execute("select kernel_name, value, duration from counters_collection where counter_name=? order by start", (counter,)).fetchall()
    finally:
        con.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("log")
    ap.add_argument("fetch")
    ap.add_argument("write")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_counter_calibration"))
    a = ap.parse_args()
    launches = [ln.split()[1:] for ln in open(a.log) if ln.startswith("CALIB ")]
    mine = lambda r: "k_gather" in r[0] or "k_stream" in r[0] or "k_scatter" in r[0] or "k_append" in r[0]
    f = [r for r in rows(a.fetch, "FETCH_SIZE") if mine(r)]
    w = [r for r in rows(a.write, "WRITE_SIZE") if mine(r)]
    assert len(f) == len(launches) == len(w), (len(f), len(w), len(launches))
    acc = {}
    for (name, table, rd, wr), fr, wrr in zip(launches, f, w):
        k = (name, int(table))
        d = acc.setdefault(k, {"read": int(rd), "write": int(wr), "fetch_kib": [], "write_kib": [], "us": []})
        d["fetch_kib"].append(fr[1])
        d["write_kib"].append(wrr[1])
        d["us"].append(fr[2] / 1e3)
    out = {"shapes": {}}
    lines = ["# FETCH_SIZE / WRITE_SIZE calibration on this engine's access shapes (MI355X, rocprofv3 --pmc, separate passes)", "",
             "`tools/counter_calib.hip`: every launch moves an exactly known number of bytes.  factor = algorithmic bytes / (counter KiB x 1024):",
             "what the counter must be multiplied by to read as bytes in that shape.  Median of 3 launches.", "",
             "| shape | table | algorithmic read MB | FETCH_SIZE MB (raw) | fetch factor | algorithmic write MB | WRITE_SIZE MB (raw) | write factor | us | algorithmic GB/s |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    for (name, table), d in acc.items():
        fk = statistics.median(d["fetch_kib"]) * 1024
        wk = statistics.median(d["write_kib"]) * 1024
        us = statistics.median(d["us"])
        ff = d["read"] / fk if d["read"] and fk else None
        wf = d["write"] / wk if d["write"] and wk else None
        out["shapes"][f"{name}@{table >> 20}MiB"] = {"read_bytes": d["read"], "write_bytes": d["write"], "fetch_counter_bytes": fk, "write_counter_bytes": wk,
                                                  "fetch_factor": ff, "write_factor": wf, "us": us}
        lines.append(f"| {name} | {table >> 20} MiB | {d['read'] / 1e6:.1f} | {fk / 1e6:.1f} | {ff:.3f} |" if ff else f"| {name} | {table >> 20} MiB | 0 | {fk / 1e6:.1f} | - |")
        lines[-1] += (f" {d['write'] / 1e6:.1f} | {wk / 1e6:.1f} | {wf:.3f} |" if wf else f" 0 | {wk / 1e6:.1f} | - |") + f" {us:.1f} | {(d['read'] + d['write']) / us / 1e3:.0f} |"
    json.dump(out, open(a.out + ".json", "w"), indent=1)
    open(a.out + ".md", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
