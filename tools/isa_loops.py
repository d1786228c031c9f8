#!/usr/bin/env python3
"""Loops of one kernel in hipcc's -S output: for every backward branch, the instruction mix of the body.
    tools/isa_loops.py produni.s nbp_product_kernel_t2_e2_xs [--dump START_LABEL]"""
import re
import sys
from collections import Counter

path, kern = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\d+" + kern + r"PK|^" + kern + ":", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".amdhsa_") or lines[i].startswith("\t.section"))
body = lines[start:end]
labels, instrs = {}, []
for l in body:
    s = l.strip()
    m = re.match(r"^(\.LBB[0-9_]+):", s)
    if m:
        labels[m.group(1)] = len(instrs)
        continue
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
        continue
    instrs.append(s.split(";")[0].strip())
print(f"{kern}: {len(instrs)} instructions")
loops = []
for i, ins in enumerate(instrs):
    m = re.match(r"s_cbranch_\w+\s+(\.LBB[0-9_]+)|s_branch\s+(\.LBB[0-9_]+)", ins)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] <= i:
            loops.append((labels[t], i, t))
def kind(op):
    if op.startswith("v_") and ("f64" in op): return "v_f64"
    if op.startswith("v_"): return "v_other"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    return "other"
for a, b, t in sorted(loops):
    c = Counter(kind(x.split()[0]) for x in instrs[a:b + 1])
    inner = [l for l in loops if a < l[0] and l[1] < b]
    print(f"  loop {t:12s} [{a:5d}..{b:5d}] n={b - a + 1:4d} {dict(c)}" + (f"  (contains {len(inner)} loops)" if inner else ""))
if len(sys.argv) > 4 and sys.argv[3] == "--dump":
    t = sys.argv[4]
    a = labels[t]
    b = max(l[1] for l in loops if l[2] == t)
    for x in instrs[a:b + 1]:
        print("      " + x)
