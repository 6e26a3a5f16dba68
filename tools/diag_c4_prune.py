#!/usr/bin/env python3
"""c4 on one GPU (N = 100k, K = 500k): per frame the tree searches of the NN / the prune, the rows scored by particle waves / off the list
(round 6: what the prune fix costs there, and what the distance field takes back; MIDAS_MESH_FIELD=0 for the form without it)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.engine import PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
K, D, N = 500_000, 512, 100_000
cb = make_codebook("025_mug", K=K, D=D, seed=1004); tr = make_trajectory(cb, T=70, seed=2004)
eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
d0 = np.linalg.norm(cb.poses[:, :3, 3] - tr.gt_poses[0][:3, 3], axis=1)
near = np.argsort(d0)[: K // 20]
eng.set_particles(torch.as_tensor(cb.poses[np.random.default_rng(4).choice(near, N)])); eng.project_to_codebook()
od, co = torch.as_tensor(tr.odoms).to(dev), torch.as_tensor(tr.codes).to(dev)
print("mesh verts", cb.mesh_vertices.shape, "extent", cb.mesh_vertices.max(0) - cb.mesh_vertices.min(0))
for i in range(10): eng.step(od[1 + i % 68], co[1 + i % 68])
t0 = eng.telemetry.cpu().numpy().copy()
n = 60
torch.cuda.synchronize(); t3 = time.perf_counter()
for i in range(n): eng.step(od[1 + (10 + i) % 68], co[1 + (10 + i) % 68])
torch.cuda.synchronize()
us = (time.perf_counter() - t3) / n * 1e6
t1 = eng.telemetry.cpu().numpy()
print("us/step", round(us, 1), "per frame: nn tree", (t1[0] - t0[0]) / n, "prune tree", (t1[1] - t0[1]) / n, "rows by particle waves", (t1[2] - t0[2]) / n, "rows off list", (t1[3] - t0[3]) / n, "kept", eng.status.cpu().numpy().tolist())
