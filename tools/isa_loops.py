#!/usr/bin/env python3
"""ISA histogram of the loops of one kernel in a hipcc -S --cuda-device-only listing: for every backward branch target
(loop), the instruction count by class.  usage: isa_loops.py file.s <mangled kernel name> [min_instrs]"""
import re
import sys
from collections import Counter

src, kern = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
ins = []
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    ins.append(t.split(";")[0].strip())


def cls(op):
    o = op.split()[0]
    if "dpp" in op and o.startswith("v_"):
        return "valu_dpp"
    if re.match(r"v_(add|mul|fma|max|min|div|rcp|fract|trig|ldexp|frexp|cvt_f64|cvt_.*_f64|cmp.*_f64|cndmask.*64|rndne_f64|floor_f64).*f64", o) or o.endswith("_f64"):
        return "valu_f64"
    if o.startswith("v_pk_"):
        return "valu_pk"
    if o.startswith("v_cmp") or o.startswith("v_cndmask"):
        return "valu_cmp_sel"
    if o.startswith("v_mov") or o.startswith("v_accvgpr"):
        return "valu_mov"
    if o.startswith("v_cvt"):
        return "valu_cvt"
    if re.match(r"v_.*_f32", o):
        return "valu_f32"
    if o.startswith("v_"):
        return "valu_int"
    if o.startswith("ds_"):
        return "lds"
    if o.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if o.startswith("s_waitcnt"):
        return "s_waitcnt"
    if o.startswith("s_barrier"):
        return "s_barrier"
    if o.startswith("s_"):
        return "salu"
    return "other"


loops = []
for i, t in enumerate(ins):
    m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", t) or re.match(r"s_branch\s+(\.LBB\d+_\d+)", t)
    if m and m.group(1) in labels and labels[m.group(1)] <= i:
        loops.append((labels[m.group(1)], i, m.group(1)))
print(f"{kern}: {len(ins)} instructions, {len(loops)} loops")
for a, b, lab in sorted(loops):
    n = b - a + 1
    if n < minn:
        continue
    c = Counter(cls(t) for t in ins[a:b + 1])
    valu = sum(v for k, v in c.items() if k.startswith("valu"))
    print(f"  loop {lab} [{a}..{b}] {n} instrs, VALU {valu}: " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
    if len(sys.argv) > 4 and sys.argv[4] == lab:
        for t in ins[a:b + 1]:
            print("      ", t)
