#!/usr/bin/env python3
"""Kernel trace csv (rocprofv3 --kernel-trace --output-format csv) -> duration of one kernel (regex) over time, in groups."""
import csv, re, sys
path, rx, group = sys.argv[1], re.compile(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 20
rows = [r for r in csv.DictReader(open(path)) if rx.search(r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(d), "launches; mean us per group of", group, ":", [round(sum(d[i:i + group]) / len(d[i:i + group]), 1) for i in range(0, len(d), group)])
