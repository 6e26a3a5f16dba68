#!/usr/bin/env python3
"""Per-frame time of the pipelined engine over the first frames after init_filter(gt_0, N) + projection (the bench's start)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.engine import PipelinedFilterEngine, FilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory, mesh_scale
from scipy.spatial.transform import Rotation
dev = torch.device("cuda", 0)
N, K, D, T = 100000, 50000, 512, 260
cb = make_codebook(K=K, D=D, seed=1001)
traj = make_trajectory(cb, T=T + 2, seed=2001)
cls = FilterEngine if len(sys.argv) > 1 and sys.argv[1] == "eager" else PipelinedFilterEngine
eng = cls(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4000, device=dev)
g = torch.Generator().manual_seed(100)
tn0 = torch.normal(0.0, mesh_scale(cb.extents) / 3.0, size=(N, 3), generator=g)
rn0 = torch.normal(0.0, 60.0, size=(N, 3), generator=g)
Tn = torch.zeros((N, 4, 4))
Tn[:, :3, :3] = torch.as_tensor(Rotation.from_euler("zyx", rn0.numpy(), degrees=True).as_matrix()).float()
Tn[:, :3, 3], Tn[:, 3, 3] = tn0, 1.0
eng.set_particles(torch.as_tensor(traj.gt_poses[0])[None] @ Tn)
eng.project_to_codebook()
od, co, gt = (torch.as_tensor(x).to(dev) for x in (traj.odoms, traj.codes, traj.gt_poses))
evs = [torch.cuda.Event(enable_timing=True) for _ in range(T + 1)]
kept, dist, rm = [], [], []
evs[0].record()
for t in range(1, T + 1):
    eng.step(od[t], co[t], gt=gt[t])
    evs[t].record()
    if t % 20 == 0 or t < 6:
        st = eng.status.cpu().numpy()  # materialises (a flush): only every 20th frame
        kept.append((t, int(st[1]), len(torch.unique(eng.nn_idx)), float(eng.rmse[0]) * 1e3))
torch.cuda.synchronize()
ms = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(T)])
print("us/frame by 20s:", [int(1e3 * ms[i:i + 20].mean()) for i in range(0, T, 20)])
print("(frame, kept, distinct NN, rmse mm):", kept)
print("telemetry", eng.telemetry.cpu().numpy()[:2])
