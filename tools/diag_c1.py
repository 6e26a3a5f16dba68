#!/usr/bin/env python3
"""c1 (N=1000, K=5000, D=256): per-frame time by step() calls and by run() calls of several lengths (GPU box only)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.engine import PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
N, K, D = int(os.environ.get("C1_N", 1000)), int(os.environ.get("C1_K", 5000)), 256
cb = make_codebook("004_sugar_box", K=K, D=D, seed=1000)
tr = make_trajectory(cb, T=262, seed=2000)
od, co, gt = (torch.as_tensor(a).to(dev) for a in (tr.odoms, tr.codes, tr.gt_poses))
eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
eng.set_particles(torch.as_tensor(cb.poses[np.random.default_rng(0).integers(0, K, N)])); eng.project_to_codebook()
def t_step(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): eng.step(od[1 + i % 250], co[1 + i % 250])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
def t_run(n, with_gt):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.run(od[1:1 + n], co[1:1 + n], gt[1:1 + n] if with_gt else None)
    t1 = time.perf_counter(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6, (t1 - t0) / n * 1e6
print("step x100: %.1f us" % t_step(100)); print("step x100: %.1f us" % t_step(100))
for n in (20, 50, 100, 200, 200):
    for g in (False, True):
        a, b = t_run(n, g)
        print("run(%d, gt=%s): %.1f us/frame, host enqueue %.1f us/frame" % (n, g, a, b))
print("telemetry", eng.telemetry.cpu().numpy()[:6])
