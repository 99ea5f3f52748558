#!/usr/bin/env python3
"""Summarise rocprofv3 runs of bench.py (rocpd sqlite output) into profiles/*.md + profiles/traffic.json.

usage: rocprof_summary.py <tag> <stats.db> [<fetch.db> <write.db>] [--workload C4] [--steps K --warmup W]

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE come from
separate --pmc passes and are in KiB.  Round 4 calibrated them on this engine's own access shapes (profiles/r04_counter_calibration.md):
FETCH_SIZE counts 64 B per read request whatever its size -- exact for 4-16 B gathers (one 64 B sector each), half for coalesced
128 B stream requests; WRITE_SIZE is exact.  So raw + write is a LOWER bound, 2 x raw + write (the guide's stream correction applied to
everything) an UPPER bound; with --items N the walk's stream share (its own frontier, read back once, + the items) gives the estimate
bench.py's calibrated_traffic() uses.
"""
import argparse
import json
import os
import sqlite3
import sys


def q(db, sql):
    con = sqlite3.connect(db)
    try:
        return con.execute(sql).fetchall()
    finally:
        con.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("stats")
    ap.add_argument("fetch", nargs="?")
    ap.add_argument("write", nargs="?")
    ap.add_argument("--workload", default="C4")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--kernel", default="k_expand")
    ap.add_argument("--items", type=int, default=0, help="requests per launch: enables the calibrated estimate (frontier bytes = WRITE_SIZE - 6 B x items)")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    lines = [f"# rocprofv3 summary `{a.tag}` ({a.workload}, bench.py --steps {a.steps})", "",
             "## --kernel-trace --stats (per kernel)", "", "| kernel | calls | total us | avg us | % |", "|---|---|---|---|---|"]
    for name, calls, total, avg, pct in q(a.stats, "select name,total_calls,total_duration,average,percentage from top_kernels"):
        short = name.replace("acl::(anonymous namespace)::", "").split("(")[0].split("<")[0]
        lines.append(f"| `{short}` | {calls} | {total:.1f} | {avg:.2f} | {pct:.2f} |")
    rows = q(a.stats, f"select start,duration,vgpr_count,sgpr_count,lds_size,grid_x,workgroup_x from kernels where name like '%{a.kernel}%' order by start")
    if rows:
        v = rows[-1]
        lines += ["", f"`{a.kernel}`: vgpr={v[2]} sgpr={v[3]} lds={v[4]} B grid={v[5]} wg={v[6]}; {len(rows)} launches, "
                  f"avg {sum(r[1] for r in rows) / len(rows) / 1e3:.2f} us", ""]
    summary = {"kernel": a.kernel, "launches": len(rows), "avg_us": sum(r[1] for r in rows) / max(1, len(rows)) / 1e3}
    per = {}
    for key, db in (("FETCH_SIZE", a.fetch), ("WRITE_SIZE", a.write)):
        if not db:
            continue
        r = q(db, f"select start,value,duration from counters_collection where counter_name='{key}' and kernel_name like '%{a.kernel}%' order by start")
        per[key] = r
        summary[key + "_KiB_total"] = sum(x[1] for x in r)
        summary[key + "_launches"] = len(r)
    if per:
        nf = len(per.get("FETCH_SIZE", []))
        nw = len(per.get("WRITE_SIZE", []))
        fk = sum(x[1] for x in per.get("FETCH_SIZE", [])) / max(1, nf)
        wk = sum(x[1] for x in per.get("WRITE_SIZE", [])) / max(1, nw)
        raw = (fk + wk) * 1024
        corr = (2 * fk + wk) * 1024
        summary.update({"fetch_bytes_per_launch_raw": fk * 1024, "write_bytes_per_launch": wk * 1024, "traffic_bytes_per_launch_raw": raw,
                        "traffic_bytes_per_launch_fetch_x2": corr})
        est = None
        if a.items:
            stream = min(max(0.0, wk * 1024 - 6.0 * a.items) + 16.0 * a.items, 2 * fk * 1024)
            est = fk * 1024 + stream / 2 + wk * 1024
            summary["traffic_bytes_per_launch_calibrated"] = est
        lines += ["## --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), kernel `%s`" % a.kernel, "",
                  f"- FETCH_SIZE: {nf} launches, mean {fk:.1f} KiB/launch", f"- WRITE_SIZE: {nw} launches, mean {wk:.1f} KiB/launch",
                  f"- HBM traffic per launch: lower bound (counters as they are: exact for gathers) {(raw) / 1e6:.2f} MB; upper bound (x2 on every read) {(corr) / 1e6:.2f} MB"
                  + (f"; calibrated estimate (stream share = the walk's own frontier + items, profiles/r04_counter_calibration.md) {est / 1e6:.2f} MB" if est else ""), ""]
        # last step, level by level
        lv = [r for r in per.get("FETCH_SIZE", [])][-7:]
        lw = [r for r in per.get("WRITE_SIZE", [])][-7:]
        lines += ["last launches (one step, level by level):", "", "| # | dur us | FETCH KiB | WRITE KiB |", "|---|---|---|---|"]
        for i, f in enumerate(lv):
            w = lw[i][1] if i < len(lw) else float("nan")
            lines.append(f"| {i + 1} | {f[2] / 1e3:.1f} | {f[1]:.0f} | {w:.0f} |")
        tj = os.path.join(a.out, "traffic.json")
        cur = json.load(open(tj)) if os.path.exists(tj) else {}
        cur[a.workload] = est if est else corr
        cur[a.workload + "_detail"] = {"tag": a.tag, **{k: v for k, v in summary.items()}}
        json.dump(cur, open(tj, "w"), indent=1)
    open(os.path.join(a.out, a.tag + ".md"), "w").write("\n".join(lines) + "\n")
    json.dump(summary, open(os.path.join(a.out, a.tag + ".json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    sys.exit(main())
