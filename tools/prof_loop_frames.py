#!/usr/bin/env python3
"""Kernel timeline of the last frames of the reference-named loop (run under rocprofv3 --kernel-trace by tools/prof_stats.sh):
usage: tools/prof_loop_frames.py run | tools/prof_loop_frames.py show <kernel_trace.csv> [frames]"""
import csv, os, sys
if sys.argv[1] == "run":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from midastouch_amd.config import load_config
    from midastouch_amd.filter import filter as run_filter, synthetic_sequence
    cfg = load_config(["expt.params.num_particles=100000", "expt.codebook_size=50000", "tcn.model.output_dim=512"])
    dev = torch.device("cuda", 0)
    seq = synthetic_sequence(cfg, dev, T=120, D=512)
    run_filter(cfg, seq, device=dev, max_frames=10)
    st = run_filter(cfg, seq, device=dev, cluster=True, draws="device", floor=1000, max_frames=120)
    print("N_final", st["num_particles"][-1], "median ms", sorted(st["time"][2:])[59] * 1e3)
else:
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[2])))
    nshow = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    # frames end with k_loop_resample
    ends = [i for i, r in enumerate(rows) if "k_loop_resample" in r[2]]
    lo = ends[-nshow - 1] + 1
    t0 = rows[lo][0]
    for s, e, n in rows[lo:ends[-1] + 1]:
        print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  {n[:90]}")
