#!/usr/bin/env python3
"""Workload for rocprofv3 --kernel-trace --stats: the reference-named loop (filter()) with clustering + annealing,
N0 particles, T frames (one setting per process so the kernel statistics belong to it)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.config import load_config
from midastouch_amd.filter import filter as run_filter, synthetic_sequence

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
floor = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
T = int(sys.argv[3]) if len(sys.argv) > 3 else 200
cfg = load_config([f"expt.params.num_particles={N}", "expt.codebook_size=50000", "tcn.model.output_dim=512"])
dev = torch.device("cuda", 0)
seq = synthetic_sequence(cfg, dev, T=T, D=512)
st = run_filter(cfg, seq, device=dev, floor=floor)
steady = sorted(st["time"][2:])
print("frames", len(st["time"]), "steady ms/frame", 1e3 * sum(steady) / len(steady), "median", 1e3 * steady[len(steady) // 2], "N", st["num_particles"][::25])
