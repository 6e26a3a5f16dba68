#!/usr/bin/env python3
"""run(T) (one C-ABI call) against T step() calls from the same state: wall and device time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.engine import PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
N, K, D, T = 100000, 50000, 512, 100
cb = make_codebook(K=K, D=D, seed=1001)
traj = make_trajectory(cb, T=2 * T + 30, seed=2001)
eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4000, device=dev)
rng = np.random.default_rng(0)
d0 = np.linalg.norm(cb.poses[:, :3, 3] - traj.gt_poses[0][:3, 3], axis=1)
near = np.argsort(d0)[:2500]
eng.set_particles(torch.as_tensor(cb.poses[rng.choice(near, N)]))
od, co, gt = (torch.as_tensor(x).to(dev) for x in (traj.odoms, traj.codes, traj.gt_poses))
for t in range(1, 21):
    eng.step(od[t], co[t], gt=gt[t])
torch.cuda.synchronize()
for name in ("step", "run", "step", "run"):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    if name == "step":
        for t in range(21, 21 + T):
            eng.step(od[t], co[t], gt=gt[t])
    else:
        eng.run(od[21:21 + T], co[21:21 + T], gt[21:21 + T])
    t_enq = time.perf_counter() - t0
    e1.record(); torch.cuda.synchronize()
    print(f"{name:5s}: enqueue {1e6 * t_enq / T:7.1f} us/frame, wall {1e6 * (time.perf_counter() - t0) / T:7.1f} us/frame, device {1e3 * e0.elapsed_time(e1) / T:7.1f} us/frame")
