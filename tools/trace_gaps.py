#!/usr/bin/env python3
"""Largest idle gaps of the device in a rocprofv3 kernel trace: usage tools/trace_gaps.py <kernel_trace.csv> [n]
(prints the n largest gaps between the end of one kernel and the start of the next, with the kernels on either side)"""
import csv, sys
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
gaps = []
end = rows[0][1]
for i in range(1, len(rows)):
    s, e, name = rows[i]
    if s > end:
        gaps.append((s - end, i))
    end = max(end, e)
gaps.sort(reverse=True)
t0 = rows[0][0]
for g, i in gaps[:n]:
    print(f"gap {g / 1e6:8.3f} ms at t = {(rows[i][0] - t0) / 1e6:9.3f} ms: after [{rows[i - 1][2][:60]}] ({(rows[i - 1][1] - rows[i - 1][0]) / 1e3:.1f} us) before [{rows[i][2][:60]}] ({(rows[i][1] - rows[i][0]) / 1e3:.1f} us)")
