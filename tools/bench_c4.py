#!/usr/bin/env python3
"""BASELINE config 4 on ONE GPU (GPU box only): N = 100k particles against the whole 500k x 512 codebook (1 GB of
embeddings; the 8-GPU form shards the rows) - the regime where the codebook stream dominates the frame."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.engine import PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
K, D, N = int(os.environ.get("C4_K", 500_000)), 512, 100_000
t0 = time.perf_counter()
cb = make_codebook("025_mug", K=K, D=D, seed=1004); tr = make_trajectory(cb, T=70, seed=2004)
t1 = time.perf_counter()
eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
torch.cuda.synchronize(); t2 = time.perf_counter()
d0 = np.linalg.norm(cb.poses[:, :3, 3] - tr.gt_poses[0][:3, 3], axis=1)
near = np.argsort(d0)[: K // 20]
eng.set_particles(torch.as_tensor(cb.poses[np.random.default_rng(4).choice(near, N)])); eng.project_to_codebook()
od, co = torch.as_tensor(tr.odoms).to(dev), torch.as_tensor(tr.codes).to(dev)
for i in range(10): eng.step(od[1 + i % 68], co[1 + i % 68])
torch.cuda.synchronize(); t3 = time.perf_counter()
n = 60
for i in range(n): eng.step(od[1 + (10 + i) % 68], co[1 + (10 + i) % 68])
torch.cuda.synchronize()
us = (time.perf_counter() - t3) / n * 1e6
bytes_step = K * (4 * D + 24) + (4 * D + 4 * K) + N * 340
print(json.dumps({"c4total_N100k_K%dk_D512_single_gpu_pipelined" % (K // 1000): {
    "us_per_step": round(us, 1), "steps_per_s": round(1e6 / us), "algorithmic_MB": round(bytes_step / 1e6, 1),
    "step_GBps": round(bytes_step / us / 1e3, 1), "step_frac_of_8TBps": round(bytes_step / us / 1e3 / 8000, 3),
    "host_codebook_s": round(t1 - t0, 1), "index_build_s": round(t2 - t1, 1)}}))
