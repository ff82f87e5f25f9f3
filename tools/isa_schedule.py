#!/usr/bin/env python3
"""Run-length view of one kernel's instruction schedule from a device-only assembly file, from `first s_setprio - back`
lines on: f = f64 VALU, i = other VALU, p = permlane swap, R / W = LDS read / write, G = global load, S = scratch,
w(...) = s_waitcnt, B = branch.   tools/isa_schedule.py file.s <mangled-name substring> [back] [length]"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 400
length = int(sys.argv[4]) if len(sys.argv) > 4 else 3400
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]


def cls(t):
    k = t.split()[0]
    if k.startswith("ds_read"):
        return "R"
    if k.startswith("ds_write"):
        return "W"
    if k.startswith("global_load"):
        return "G"
    if k.startswith("scratch_"):
        return "S"
    if k == "s_waitcnt":
        return "w(%s)" % (" ".join(t.split()[1:]))
    if re.match(r"v_(fma|fmac|add|mul|rndne)_f64|v_cvt_f64", k):
        return "f"
    if k.startswith("v_permlane"):
        return "p"
    if k.startswith("v_accvgpr"):
        return "A"
    if k.startswith("v_"):
        return "i"
    if k == "s_setprio":
        return "\nPRIO%s " % t.split()[1]
    if k.startswith("s_cbranch") or k == "s_branch":
        return "B"
    if k == "s_sleep":
        return "SLEEP"
    return ""


first = next(i for i, l in enumerate(body) if "s_setprio" in l)
out, prev, cnt = [], None, 0
for l in body[max(0, first - back):first + length]:
    t = l.strip()
    if l.startswith(".LBB"):
        c = "\n[" + l.split(":")[0] + "]"
    elif not t or t.startswith(";") or t.startswith("."):
        continue
    else:
        c = cls(t)
    if not c:
        continue
    if c == prev and len(c) == 1:
        cnt += 1
    else:
        if prev:
            out.append(prev + (str(cnt) if cnt > 1 else ""))
        prev, cnt = c, 1
out.append(prev + (str(cnt) if cnt > 1 else ""))
print(" ".join(out))
