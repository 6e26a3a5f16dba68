#!/usr/bin/env python3
"""Per kernel of a .s file: global loads, vmcnt waits, vmcnt(0) waits - spots load chains the compiler serialised."""
import re, sys
cur, stats = None, {}
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1); stats[cur] = [0, 0, 0]
    if cur is None: continue
    if "global_load" in line or "buffer_load" in line: stats[cur][0] += 1
    if "s_waitcnt" in line and "vmcnt" in line:
        stats[cur][1] += 1
        if "vmcnt(0)" in line: stats[cur][2] += 1
for k, (l, w, w0) in stats.items():
    if l: print(f"{l:5d} loads {w:5d} waits {w0:5d} vmcnt(0)  {k[:70]}")
