#!/usr/bin/env python3
"""Per-call durations of the front kernel from a rocprofv3 kernel trace (p_kernel_trace.csv): mean over all calls (what
kernel_stats.csv reports), median, and the mean over the calls of the bench's timed region and per-step pass (the last
2 x steps calls).  usage: tools/front_trace_stats.py <p_kernel_trace.csv> <steps> [out.json]"""
import csv, json, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = {}
for pat, key in (("k_frame_front<float, 8, 2,", "frame_front_pipelined"), ("k_tail_a3", "tail_a3"), ("k_tail_a2d", "tail_a2d")):
    d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if pat in r["Kernel_Name"]])
    if len(d) == 0:
        continue
    t = d[-2 * steps:]
    out[key] = {"calls": int(len(d)), "mean_us_all_calls": float(d.mean()), "median_us_all_calls": float(np.median(d)),
                "timed_region_calls": int(len(t)), "mean_us_timed_region": float(t.mean()), "median_us_timed_region": float(np.median(t)),
                "p95_us_timed_region": float(np.percentile(t, 95))}
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
