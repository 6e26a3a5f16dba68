#!/usr/bin/env python3
"""The last frames of a rocprofv3 kernel trace of the loop (tools/prof_loop.py): start / end / duration of every kernel.
usage: tools/loop_tail_timeline.py <dir with *kernel_trace.csv> [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48]) for r in csv.DictReader(open(f))]
rows = sorted(r for r in rows if "midas" in r[2])
sel = rows[-n - 2:-2]
t0 = sel[0][0]
for s, e, k in sel:
    print(f"{(s - t0) / 1e3:9.2f} {(e - t0) / 1e3:9.2f}  dur {(e - s) / 1e3:7.2f}  {k}")
res = [r for r in rows if "k_loop_resample" in r[2]][-60:]
print("frame period over the last %d frames: %.1f us" % (len(res) - 1, (res[-1][0] - res[0][0]) / 1e3 / (len(res) - 1)))
