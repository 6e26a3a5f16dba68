#!/usr/bin/env python3
"""Print start/end (us, relative) of the kernels of a few frames from a rocprofv3 kernel_trace.csv."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r.get("Stream_Id", r.get("Queue_Id", "")))
        for r in csv.DictReader(open(f))]
rows = [r for r in rows if 'midas' in r[2]]
rows.sort()
# take frames from the middle
mid = len(rows) // 2
sel = rows[mid: mid + int(sys.argv[2]) if len(sys.argv) > 2 else mid + 16]
t0 = sel[0][0]
for s, e, n, q in sel:
    print(f"{(s - t0) / 1e3:9.2f} {(e - t0) / 1e3:9.2f}  dur {(e - s) / 1e3:7.2f}  q{q}  {n}")
