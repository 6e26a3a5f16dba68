#!/usr/bin/env python3
"""Frame rate of the reference-shaped loop (midastouch_amd.filter.filter: the class-surface calls, one op at a time,
with the per-frame synchronisations the reference has) next to the fused FilterEngine."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.config import load_config
from midastouch_amd.filter import filter as run_filter, synthetic_sequence

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
cfg = load_config([f"expt.params.num_particles={N}", f"expt.codebook_size={K}", "tcn.model.output_dim=512"])
dev = torch.device("cuda", 0)
seq = synthetic_sequence(cfg, dev, T=80, D=512)
for cluster in (False, True):
    st = run_filter(cfg, seq, device=dev, max_frames=60, cluster=cluster)
    t = st["time"][10:]
    print(f"cluster={cluster}: {1e3 * sum(t) / len(t):.2f} ms/frame over {len(t)} frames, final N={st['num_particles'][-1]}, "
          f"rmse_t={1e3 * st['rmse_t'][-1]:.2f} mm")
