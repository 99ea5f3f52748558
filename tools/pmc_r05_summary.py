#!/usr/bin/env python3
"""usage: tools/pmc_r05_summary.py <gpurun_out/prof/r05_pmc_CFG> <CFG> -- markdown summary of the per-set rocprofv3 --pmc passes of tools/pmc_r05.sh:
mean per launch of k_check_local over the launches of every pass, plus the ratios the DESIGN text quotes."""
import glob
import os
import sqlite3
import sys

root, cfg = sys.argv[1], sys.argv[2]
rnd = sys.argv[3] if len(sys.argv) > 3 else "round-5"
tool = sys.argv[4] if len(sys.argv) > 4 else "tools/pmc_r05.sh"
vals, durs, fails = {}, [], []
for d in sorted(glob.glob(os.path.join(root, "s*"))):
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    if not dbs:
        fails.append(os.path.basename(d) + ": " + (open(os.path.join(d, "run.log")).read()[-200:].replace("\n", " ") if os.path.exists(os.path.join(d, "run.log")) else "no log"))
        continue
    con = sqlite3.connect(dbs[0])
    rows = con.execute("select counter_name, value, duration from counters_collection where kernel_name like '%k_check_local%'").fetchall()
    con.close()
    by = {}
    for c, v, dur in rows:
        by.setdefault(c, []).append(v)
        durs.append(dur / 1e3)
    for c, v in by.items():
        vals[c] = (sum(v) / len(v), len(v))
print(f"# rocprofv3 --pmc passes on the {rnd} `k_check_local`, {cfg} (262 144-item batch; {tool})\n")
print("One pass per counter set (own run, `--kernel-trace --pmc` only), mean per launch over the launches of the pass; counters with `_sum` are summed over the")
print(f"instances (XCDs / channels) by rocprofv3.  Kernel duration under the profiler: {sum(durs) / max(1, len(durs)):.1f} us (mean over {len(durs)} samples; counter collection serialises launches).\n")
print("| counter | per launch | launches |\n|---|---|---|")
for c in sorted(vals):
    print(f"| `{c}` | {vals[c][0]:,.0f} | {vals[c][1]} |")
g = lambda k: vals.get(k, (None, 0))[0]  # noqa: E731
print("\n## derived\n")
n = 262144
if g("SQ_INSTS_VMEM_RD"):
    print(f"* vector-memory read instructions per Check: {g('SQ_INSTS_VMEM_RD') / n:.1f} wave-instructions (x 64 lanes); writes: {(g('SQ_INSTS_VMEM_WR') or 0) / n:.2f}; LDS: {(g('SQ_INSTS_LDS') or 0) / n:.1f}; VALU: {(g('SQ_INSTS_VALU') or 0) / n:.0f}; SALU: {(g('SQ_INSTS_SALU') or 0) / n:.0f}")
    if g("SQ_INSTS_LDS"):
        print(f"* LDS : VMEM-read instruction ratio {g('SQ_INSTS_LDS') / g('SQ_INSTS_VMEM_RD'):.2f}")
if g("SQ_WAVE_CYCLES") and g("SQ_BUSY_CYCLES"):
    print(f"* `SQ_WAVE_CYCLES` / `SQ_BUSY_CYCLES` = {g('SQ_WAVE_CYCLES') / g('SQ_BUSY_CYCLES'):.2f} (waves resident per busy SQ cycle, as the counter is scaled)")
    if g("SQ_WAIT_INST_ANY"):
        print(f"* waves waiting on ANY outstanding instruction: `SQ_WAIT_INST_ANY` / `SQ_WAVE_CYCLES` = {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.2f} of the wave-cycles")
    for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
        if g(k):
            print(f"* `{k}` / `SQ_BUSY_CYCLES` = {g(k) / g('SQ_BUSY_CYCLES'):.3f}")
if g("TCP_TOTAL_CACHE_ACCESSES_sum"):
    print(f"* vector-L1 accesses per Check: {g('TCP_TOTAL_CACHE_ACCESSES_sum') / n:.1f}; of which go on to the L2 (`TCP_TCC_READ_REQ`): {(g('TCP_TCC_READ_REQ_sum') or 0) / n:.1f} "
          f"({100.0 * (g('TCP_TCC_READ_REQ_sum') or 0) / g('TCP_TOTAL_CACHE_ACCESSES_sum'):.0f} %)")
if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and (g("TCC_HIT_sum") + g("TCC_MISS_sum")) > 0:
    print(f"* L2: hit rate {100.0 * g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.1f} % ({g('TCC_HIT_sum') / n:.1f} hits, {g('TCC_MISS_sum') / n:.1f} misses per Check)")
if g("FETCH_SIZE") and g("WRITE_SIZE"):
    print(f"* `FETCH_SIZE` {g('FETCH_SIZE') * 1024 / 1e6:.1f} MB, `WRITE_SIZE` {g('WRITE_SIZE') * 1024 / 1e6:.1f} MB per launch (raw counters, KiB -> bytes; calibration: profiles/r04_counter_calibration.md)")
for f in fails:
    print(f"\n(pass failed: {f})")
