#!/usr/bin/env python3
"""Kernel durations inside the TIMED REGION of `bench.py --steps K --warmup W` from a rocprofv3 kernel trace of that command
(p_kernel_trace.csv).  bench.py's frames in launch order: 2 first-touch frames, 20 diffuse-regime frames (unless --no-diffuse),
W warm-up frames, K timed frames, ...; every frame launches one k_frame_front* and one k_tail_a3 (k_tail_a2d with MIDAS_TAIL_GROUPED=0).
usage: tools/driver_trace_stats.py <p_kernel_trace.csv> <warmup> <steps> [--no-diffuse] [out.json]"""
import csv, json, sys
import numpy as np
args = [a for a in sys.argv[1:] if not a.startswith("--")]
rows = list(csv.DictReader(open(args[0])))
W, K = int(args[1]), int(args[2])
skip = 2 + (0 if "--no-diffuse" in sys.argv else 20) + W
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = {"command": f"bench.py --steps {K} --warmup {W}", "frames_before_timed_region": skip}
for pat, key in (("k_frame_front", "frame_front"), ("k_tail_a", "tail_a")):
    sel = [r for r in rows if pat in r["Kernel_Name"]]
    d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sel])
    t = d[skip:skip + K]
    out[key] = {"launches_total": int(len(d)), "timed_region_launches": int(len(t)), "mean_us_timed_region": float(t.mean()),
                "median_us_timed_region": float(np.median(t)), "max_us_timed_region": float(t.max()),
                "first_us": [round(float(x), 1) for x in t[:5]], "last_us": [round(float(x), 1) for x in t[-5:]], "all_us": [round(float(x), 1) for x in t],
                "kernel_names_timed_region": sorted({r["Kernel_Name"].split("(")[0][:80] for r in sel[skip:skip + K]})}
fr = [r for r in rows if "k_frame_front" in r["Kernel_Name"]][skip:skip + K]
ta = [r for r in rows if "k_tail_a" in r["Kernel_Name"]][skip:skip + K]
if len(fr) == K and len(ta) == K:
    span = (int(ta[-1]["End_Timestamp"]) - int(fr[0]["Start_Timestamp"])) / 1e3
    out["timed_region_device_span_us"] = span
    out["timed_region_us_per_frame"] = span / K
print(json.dumps(out, indent=1))
if len(args) > 3:
    json.dump(out, open(args[3], "w"), indent=1)
