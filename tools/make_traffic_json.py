#!/usr/bin/env python3
"""profiles/r02_traffic.json from a tools/pmc_traffic.sh output directory.

usage: tools/make_traffic_json.py gpurun_out/pmc_<tag> [out.json]
HBM bytes per launch: fetch = FETCH_SIZE (KiB) * 1024 * 2 (gfx950's rocprofv3 tallies 128-B requests at 64 B,
MI355X_MICROARCH.md "HBM"; cross-checked against TCC_EA0_RDREQ * 128 B), write = WRITE_SIZE (KiB) * 1024.
"""
import csv, glob, json, os, sys
from collections import defaultdict

d = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_traffic.json")
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
GROUPS = {"k_frame_front<float, 8, 2,": "frame_front", "k_frame_front<float, 8, 1,": "frame_front", "k_frame_front<float, 8, 0,": "frame_front_plain",
          "8, true>": "frame_front", "8, false>": "frame_front_plain", "k_tail_a3": "tail_a", "k_tail_a2": "tail_a", "k_tail_b2": "tail_b", "k_score_reg<float, 8, 0>": "score_codebook",
          "k_particle_update": "particle_update", "k_tail_a(": "tail_a_legacy", "k_tail_b(": "tail_b_legacy"}
res = {}
for kname, cs in acc.items():
    key = next((g for pat, g in GROUPS.items() if pat in kname), None)
    if key is None or "FETCH_SIZE" not in cs or len(cs["FETCH_SIZE"]) < 5:
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    fetch, write = m["FETCH_SIZE"] * 1024 * 2, m.get("WRITE_SIZE", 0.0) * 1024
    res[key] = {"fetch_bytes": fetch, "write_bytes": write, "hbm_bytes": fetch + write,
                "tcc_ea0_rdreq_x128": m.get("TCC_EA0_RDREQ_sum", 0.0) * 128,
                "l2_hit_rate": m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) if "TCC_HIT_sum" in m else None,
                "launches": len(cs["FETCH_SIZE"])}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_traffic.sh), bench.py c2, per launch; "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B; TCC_EA0_RDREQ*128 B agrees)",
           "kernels": res}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
