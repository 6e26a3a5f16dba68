#!/usr/bin/env python3
"""Small driver for rocprofv3: the standalone NN kernel (no hint / real hints) on a realistic particle state."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import ops
from midastouch_amd.engine import FilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory

dev = torch.device("cuda", 0)
N, K, D = 100_000, 50_000, 512
cb = make_codebook(K=K, D=D, seed=1001)
traj = make_trajectory(cb, T=32, seed=2001)
eng = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
rng = np.random.default_rng(100)
d0 = np.linalg.norm(cb.poses[:, :3, 3] - traj.gt_poses[0][:3, 3], axis=1)
near = np.argsort(d0)[: max(64, K // 20)]
eng.set_particles(torch.as_tensor(cb.poses[rng.choice(near, N)]))
eng.project_to_codebook()
odoms, codes = torch.as_tensor(traj.odoms).to(dev), torch.as_tensor(traj.codes).to(dev)
for t in range(1, 12):
    eng.step(odoms[t], codes[t])
feat = ops.se3_feature(eng.poses_prop)
hint = eng.hint_next.clone()
torch.cuda.synchronize()
ops.nn6(eng.tree6, feat, None)
ops.nn6(eng.tree6, feat, hint)
ops.nn3_dist(eng.tree3, eng.poses_prop)
torch.cuda.synchronize()
