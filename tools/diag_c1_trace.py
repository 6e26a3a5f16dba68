#!/usr/bin/env python3
"""c1 with the long trajectory: 400 steps (GPU box only; run under rocprofv3 --kernel-trace)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.engine import PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
T = 262
cb = make_codebook(K=5000, D=256, seed=1000); tr = make_trajectory(cb, T=T, seed=2000)
od, co, gt = (torch.as_tensor(a).to(dev) for a in (tr.odoms, tr.codes, tr.gt_poses))
eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, 1000, device=dev)
eng.set_particles(torch.as_tensor(cb.poses[np.random.default_rng(0).integers(0, 5000, 1000)])); eng.project_to_codebook()
rows = []
for i in range(400):
    t = 1 + i % (T - 2)
    eng.step(od[t], co[t], gt=gt[t])
    if i % 20 == 19:
        torch.cuda.synchronize()
        rows.append((i, t, int(eng.status[1]), float(eng.rmse[0]) * 1e3))
for r in rows: print("frame %d traj %d valid %d rmse_t %.2f mm" % r)
