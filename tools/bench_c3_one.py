#!/usr/bin/env python3
"""c3 at its total size on one GPU (N = 1 M particles, 50k x 512 codebook), pipelined engine, 40 frames by one run() call: the command
the kernel trace and PMC passes of the c3 front profile (GPU box only)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.engine import PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
N, K, D = int(os.environ.get("C3_N", 1_000_000)), 50_000, 512
cb = make_codebook("035_power_drill", K=K, D=D, seed=1003)
tr = make_trajectory(cb, T=72, seed=2003)
eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
d0 = np.linalg.norm(cb.poses[:, :3, 3] - tr.gt_poses[0][:3, 3], axis=1)
eng.set_particles(torch.as_tensor(cb.poses[np.random.default_rng(4).choice(np.argsort(d0)[: K // 20], N)]))
eng.project_to_codebook()
od, co = torch.as_tensor(tr.odoms).to(dev), torch.as_tensor(tr.codes).to(dev)
eng.run(od[1:21], co[1:21])
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.run(od[21:61], co[21:61])
torch.cuda.synchronize()
print("c3 total on one GPU (N = %d): %.1f us per frame" % (N, (time.perf_counter() - t0) / 40 * 1e6))
