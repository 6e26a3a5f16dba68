#!/usr/bin/env python3
"""Frame rate of the reference-named loop (midastouch_amd.filter.filter: the reference's call sequence with DBSCAN every
50th frame, cluster centres and annealing every frame, N0 particles at the start) - one LoopEngine.step per frame.
Prints one JSON line per setting; frames/s = frames / (wall-clock from the first enqueue to the last frame's completion)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.config import load_config
from midastouch_amd.filter import filter as run_filter, synthetic_sequence

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
T = int(sys.argv[3]) if len(sys.argv) > 3 else 300
cfg = load_config([f"expt.params.num_particles={N}", f"expt.codebook_size={K}", "tcn.model.output_dim=512"])
dev = torch.device("cuda", 0)
seq = synthetic_sequence(cfg, dev, T=T, D=512)
run_filter(cfg, seq, device=dev, max_frames=20)  # warm-up (library load, allocator)
for cluster, draws, floor in ((True, "device", 1000), (True, "device", 100000), (False, "device", 1000), (True, "host", 1000), (True, "seeded", 1000)):
    torch.cuda.synchronize()
    t0 = time.time()
    st = run_filter(cfg, seq, device=dev, cluster=cluster, draws=draws, floor=floor, max_frames=T if draws == "device" else 60)
    wall = time.time() - t0
    n = len(st["time"])
    steady = st["time"][2:]  # without the two initial frames (host-side init_filter, filter.py:156-160)
    print(json.dumps({"N0": N, "K": K, "cluster": cluster, "draws": draws, "floor": floor, "frames": n,
                      "frames_per_s_wall_incl_init": round(n / wall, 1),
                      "frames_per_s_steady": round(len(steady) / sum(steady), 1),
                      "ms_per_frame_steady": round(1e3 * sum(steady) / len(steady), 4),
                      "ms_frame_max_steady": round(1e3 * max(steady), 3),
                      "slowest_frames": sorted(((round(1e3 * t, 3), i + 2) for i, t in enumerate(steady)), reverse=True)[:4],
                      "ms_per_frame_median": round(1e3 * sorted(steady)[len(steady) // 2], 4),
                      "host_enqueue_ms": round(1e3 * st["avg_timer"]["host_enqueue"], 4),
                      "N_final": st["num_particles"][-1], "N_min": min(st["num_particles"]),
                      "rmse_t_mm_final": round(1e3 * st["rmse_t"][-1], 2)}))
