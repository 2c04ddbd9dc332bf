"""Attributes the vector instructions of one kernel to source lines: reads a listing compiled with
`hipcc -gline-tables-only --cuda-device-only -S` (its .loc directives) and prints the lines that own the
most instructions, per loop depth range.

usage: isa_lines.py listing.s kernel-name-prefix [min-depth [max-depth [top]]]
"""
import collections
import re
import sys

text = open(sys.argv[1]).read().split("\n")
needle = sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 99
top = int(sys.argv[5]) if len(sys.argv) > 5 else 40
files, out, on = {}, [], False
for line in text:
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    if not on and re.match(r"^" + re.escape(needle), line):
        on = True
    if on:
        out.append(line)
        if "s_endpgm" in line:
            break
depth, loc = 0, None
hist = collections.Counter()
for line in out:
    m = re.match(r"^\.L(BB\d+_\d+):\s*;?(.*)", line)
    if m:
        d = re.search(r"Depth=(\d+)", m.group(2))
        depth = int(d.group(1)) if d else 0
        continue
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    if re.match(r"^\s+v_", line) and lo <= depth <= hi:
        hist[loc] += 1
print(f"vector instructions at loop depth {lo}..{hi}: {sum(hist.values())}")
for k, v in hist.most_common(top):
    print(f"  {k[0]}:{k[1]}  {v}")
