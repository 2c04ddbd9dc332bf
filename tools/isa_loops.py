"""Per-loop instruction and SGPR-spill counts of one kernel in a hipcc -S listing.

usage: isa_loops.py listing.s kernel-name-substring
"""
import collections
import re
import sys


def main(path, needle):
    text = open(path).read().split("\n")
    out, on = [], False
    for line in text:
        if not on and re.match(r"^_Z\w+:", line) and needle in line:
            on = True
        if on:
            out.append(line)
            if "s_endpgm" in line:
                break
    depth, hdr = 0, None
    total, spills, ops = collections.Counter(), collections.Counter(), collections.Counter()
    for line in out:
        m = re.match(r"^\.L(BB\d+_\d+):\s*;?(.*)", line)
        if m:
            c = m.group(2)
            d = re.search(r"Depth=(\d+)", c)
            h = re.search(r"Header: (BB\d+_\d+)", c)
            depth = int(d.group(1)) if d else 0
            hdr = h.group(1) if h else (m.group(1) if "Loop Header" in c else None)
            continue
        m = re.match(r"^\s+([vs]_\w+|ds_\w+|global_\w+|scratch_\w+|buffer_\w+|flat_\w+)", line)
        if m:
            total[(depth, hdr)] += 1
            ops[m.group(1)] += 1
            if "readlane" in m.group(1) or "writelane" in m.group(1):
                spills[(depth, hdr)] += 1
    print(f"{len(out)} lines, {sum(total.values())} instructions, spill moves {sum(spills.values())}")
    for k, v in sorted(total.items(), key=lambda x: -x[1])[:12]:
        print(f"  depth {k[0]} header {k[1]}: {v} instr, {spills[k]} readlane/writelane")
    print("  top ops:", ", ".join(f"{o} {c}" for o, c in ops.most_common(14)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
