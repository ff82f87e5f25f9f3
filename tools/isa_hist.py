#!/usr/bin/env python3
"""Instruction histogram of one kernel of a device-only assembly file (hipcc --cuda-device-only -S):
   tools/isa_hist.py file.s <substring of the mangled name> [--loop]   (--loop: only the blocks the assembler marks as inside a loop)"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
loop_only = "--loop" in sys.argv
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith("E") or (l.startswith("_Z") and key in l and ":" in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
hist = collections.Counter()
in_loop = False
for l in lines[start + 1:end]:
    if l.startswith(".LBB"):
        in_loop = "in Loop" in l or "Loop Header" in l
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    if loop_only and not in_loop:
        continue
    hist[t.split()[0]] += 1
groups = collections.Counter()
for k, v in hist.items():
    if re.match(r"v_(fma|fmac|add|mul|rndne|cvt_f64|cvt_i32_f64|fract|min|max)_f64|v_cvt_f64", k):
        groups["f64"] += v
    elif k.startswith("v_permlane"):
        groups["permlane"] += v
    elif k.startswith("v_"):
        groups["valu_other"] += v
    elif k.startswith("ds_"):
        groups["lds"] += v
    elif k.startswith(("global_", "buffer_", "scratch_", "flat_")):
        groups["vmem"] += v
    elif k.startswith("s_waitcnt"):
        groups["waitcnt"] += v
    else:
        groups["salu"] += v
print(dict(groups))
print(", ".join(f"{k} {v}" for k, v in hist.most_common(40)))
