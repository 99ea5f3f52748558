#!/usr/bin/env python3
"""usage: tools/isa_loops.py [-DNAME=VALUE ...] -- instruction mix of the hot loops of k_check_local<true, 12, false> (the wide monotone walk) in the gfx950 ISA.

Cross-compiles kernels.hip to assembly (no GPU needed), cuts the kernel out, and for every innermost loop that holds the simple expansion's three
bucket gathers (global_load_dwordx4 x 3 per step: simple_steps) prints VALU / SALU / LDS / VMEM / wait counts, the quarter-rate multiplies
(v_mul_lo_u32 / v_mul_hi_u32), 64-bit shifts and the lane spills (v_writelane / v_readlane) inside it -- what a step of 192 children costs a wave.
The walk is instruction-issue bound (profiles/r05_pmc_c4.md); this is the static half of that picture, the PMC passes are the dynamic one."""
import os
import re
import subprocess
import sys

R = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spicedb-kubeapi-proxy_amd")
out = "/tmp/isa_loops.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", *sys.argv[1:], "-o", out, f"{R}/csrc/kernels.hip"],
                      stderr=subprocess.DEVNULL)
L = open(out).read().split("\n")
want = os.environ.get("ISA_KERNEL", "k_check_localILb1ELi12ELb0E")
start = next(i for i, l in enumerate(L) if re.match(r"^_ZN.*" + want + r".*:\s", l))
end = next(i for i in range(start + 1, len(L)) if L[i].startswith("\t.section") or L[i].startswith(".Lfunc_end"))
K = L[start:end]


def cat(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    return "other"


def stats(body):
    c = {}
    for b in body:
        c[cat(b)] = c.get(cat(b), 0) + 1
    c["qmul"] = sum(b.startswith(("v_mul_lo_u32", "v_mul_hi_u32")) for b in body)
    c["shift64"] = sum(b.startswith(("v_lshrrev_b64", "v_lshlrev_b64")) for b in body)
    c["lanespill"] = sum(b.startswith(("v_writelane", "v_readlane")) for b in body)
    c["gld4"] = sum(b.startswith("global_load_dwordx4") for b in body)
    return c


ins = lambda ls: [l.split()[0] for l in ls if l.startswith("\t") and not l.strip().startswith((";", "."))]  # noqa: E731
print(f"{want}: {len(ins(K))} instructions, whole kernel {stats(ins(K))}")
hdr = [i for i, l in enumerate(K) if "Inner Loop Header" in l]
for h in hdr:
    # the label of the loop is on this line or the previous one
    m, b = None, h  # (the label sits on the first line of the header comment: a few lines up when the loop has parents)
    while b >= 0 and not m and h - b < 12:
        m = re.match(r"^(\.LBB\d+_\d+):", K[b])
        b -= 1
    if not m:
        continue
    name = m.group(1)[2:]
    idx = [j for j, l in enumerate(K) if f"Header={name} " in l or f"Header={name}\t" in l]
    lo, hi = h, (max(idx) if idx else h)
    k = hi + 1
    while k < len(K) and not K[k].startswith(".LBB"):
        k += 1
    body = ins(K[lo:k])
    st = stats(body)
    if st["gld4"] >= 3:
        print(f"  loop {name} (kernel line {lo}): {len(body)} instructions {st}")
