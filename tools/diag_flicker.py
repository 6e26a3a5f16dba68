#!/usr/bin/env python3
"""After the wide start of bench.py: how many of the codebook rows a frame needs were needed one / two / three frames before
(the prediction list holds the rows of the frame before; rows that come back after a frame's absence are claimed again)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.engine import PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory, wide_start
dev = torch.device("cuda", 0)
N, K, D = 100_000, 50_000, 512
cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
traj = make_trajectory(cb, T=140, seed=2001)
eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4000, device=dev)
eng.set_particles(torch.as_tensor(wide_start(cb.extents, traj.gt_poses[0], N, 100)))
eng.project_to_codebook()
odoms, codes, gts = (torch.as_tensor(x).to(dev) for x in (traj.odoms, traj.codes, traj.gt_poses))
hist = []
print("frame rows new_vs_1 new_vs_2 new_vs_3")
for t in range(1, 40):
    eng.step(odoms[t], codes[t], gt=gts[t])
    eng.flush()
    rows = set(np.unique(eng._nn[eng._cur].cpu().numpy()[:N]).tolist())
    if len(hist) >= 3:
        a = rows - hist[-1]
        b = a - hist[-2]
        c = b - hist[-3]
        print("%4d %6d %6d %6d %6d" % (t, len(rows), len(a), len(b), len(c)))
    hist.append(rows)
