#!/usr/bin/env python3
"""Host time to enqueue one fused step (Python + ctypes + 3 launches) next to the GPU time per step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.engine import FilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
for N, K, D in ((100_000, 50_000, 512), (1000, 5000, 256)):
    cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001); tr = make_trajectory(cb, T=130, seed=2001)
    eng = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
    rng = np.random.default_rng(0)
    d0 = np.linalg.norm(cb.poses[:, :3, 3] - tr.gt_poses[0][:3, 3], axis=1)
    eng.set_particles(torch.as_tensor(cb.poses[rng.choice(np.argsort(d0)[: max(64, K // 20)], N)])); eng.project_to_codebook()
    od, co = torch.as_tensor(tr.odoms).to(dev), torch.as_tensor(tr.codes).to(dev)
    ods, cos = [od[i] for i in range(130)], [co[i] for i in range(130)]
    for i in range(20): eng.step(ods[1 + i % 128], cos[1 + i % 128])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(300): eng.step(ods[1 + i % 128], cos[1 + i % 128])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"N={N}: enqueue {1e6 * (t1 - t0) / 300:.1f} us/step, end-to-end {1e6 * (t2 - t0) / 300:.1f} us/step")
