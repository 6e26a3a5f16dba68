#!/usr/bin/env python3
"""c5: how many distinct nearest entries does a trajectory's particle set sit on, frame by frame (GPU box only)?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.engine import PipelinedBatchFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
cb = make_codebook("cotter-pin", K=50000, D=512, seed=1005)
B, N = 64, 10000
trs = [make_trajectory(cb, T=40, seed=2200 + b) for b in range(8)]
od = torch.as_tensor(np.stack([trs[b % 8].odoms for b in range(B)], axis=1)).to(dev)
co = torch.as_tensor(np.stack([trs[b % 8].codes for b in range(B)], axis=1)).to(dev)
eng = PipelinedBatchFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, B, N, device=dev)
rng = np.random.default_rng(1)
for init in ("spread", "near"):
    if init == "spread":
        start = np.stack([cb.poses[rng.integers(0, 50000, N)] for _ in range(B)])
    else:
        start = []
        for b in range(B):
            d0 = np.linalg.norm(cb.poses[:, :3, 3] - trs[b % 8].gt_poses[0][:3, 3], axis=1)
            start.append(cb.poses[rng.choice(np.argsort(d0)[:2500], N)])
        start = np.stack(start)
    eng.set_particles(torch.as_tensor(start)); eng.project_to_codebook()
    for i in range(70):
        eng.step(od[1 + i % 38], co[1 + i % 38])
        if i in (0, 5, 10, 20, 40, 69):
            nn = eng.nn_idx.cpu().numpy()
            d = [len(np.unique(nn[b])) for b in range(B)]
            h = eng.hint.cpu().numpy()  # hints of the resampled set = what the next frame's waves start from
            dh = [len(np.unique(h[b])) for b in range(B)]
            # distinct hints per 64 consecutive slots as they are now, and if the slots were sorted by hint
            per_wave = np.mean([len(np.unique(h[0][k:k + 64])) for k in range(0, N - 63, 64)])
            hs = np.sort(h[0]); per_wave_sorted = np.mean([len(np.unique(hs[k:k + 64])) for k in range(0, N - 63, 64)])
            print(f"{init} frame {i}: distinct nn per trajectory mean {np.mean(d):.0f} (min {min(d)}, max {max(d)}); distinct hints {np.mean(dh):.0f}; "
                  f"per wave of 64 slots: {per_wave:.1f} as stored, {per_wave_sorted:.1f} sorted; valid {int(eng.status[0,1])}")
