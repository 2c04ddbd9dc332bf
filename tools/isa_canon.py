"""Counts, per loop depth of one kernel in a hipcc -S listing, the `v_max_f32 x, x, x` instructions
that re-quiet a value in front of a minimum or maximum (the compiler adds them where it cannot see that
a value is no signalling NaN: across branches and loop back-edges) and what consumes them.

usage: isa_canon.py listing.s kernel-name-prefix [min-depth]
"""
import collections
import re
import sys

text = open(sys.argv[1]).read().split("\n")
needle = sys.argv[2]
min_depth = int(sys.argv[3]) if len(sys.argv) > 3 else 0
out, on = [], False
for line in text:
    if not on and re.match(r"^" + re.escape(needle), line):
        on = True
    if on:
        out.append(line)
        if "s_endpgm" in line:
            break
depth = 0
tot, canon, follow = collections.Counter(), collections.Counter(), collections.Counter()
for i, line in enumerate(out):
    m = re.match(r"^\.L(BB\d+_\d+):\s*;?(.*)", line)
    if m:
        d = re.search(r"Depth=(\d+)", m.group(2))
        depth = int(d.group(1)) if d else 0
        continue
    m = re.match(r"^\s+(v_\w+)\s+(.*)", line)
    if not m:
        continue
    tot[depth] += 1
    if m.group(1).startswith("v_max_f32"):
        ops = [o.strip() for o in m.group(2).split(",")]
        if len(ops) == 3 and ops[1] == ops[2]:
            canon[depth] += 1
            if depth >= min_depth:
                for j in range(i + 1, min(i + 14, len(out))):
                    if re.search(r"\b" + re.escape(ops[0]) + r"\b", out[j]) and not out[j].strip().startswith(";"):
                        mm = re.match(r"^\s+(\w+)", out[j])
                        follow[mm.group(1) if mm else "?"] += 1
                        break
print("vector instructions per loop depth:", dict(sorted(tot.items())))
print("re-quieting v_max x, x per loop depth:", dict(sorted(canon.items())))
print("consumed by:", follow.most_common(8))
