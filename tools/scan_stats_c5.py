#!/usr/bin/env python3
"""Per-wave phase clocks of the batch step's particle update on c5 (run with MIDAS_ABLATE=4)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.engine import BatchFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
cb = make_codebook("cotter-pin", K=50000, D=512, seed=1005)
B, N = 64, 10000
trs = [make_trajectory(cb, T=40, seed=2200 + b) for b in range(8)]
od = torch.as_tensor(np.stack([trs[b % 8].odoms for b in range(B)], axis=1)).to(dev)
co = torch.as_tensor(np.stack([trs[b % 8].codes for b in range(B)], axis=1)).to(dev)
eng = BatchFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, B, N, device=dev)
rng = np.random.default_rng(1)
start = []
for b in range(B):
    d0 = np.linalg.norm(cb.poses[:, :3, 3] - trs[b % 8].gt_poses[0][:3, 3], axis=1)
    start.append(cb.poses[rng.choice(np.argsort(d0)[:2500], N)])
eng.set_particles(torch.as_tensor(np.stack(start))); eng.project_to_codebook()
for i in range(30): eng.step(od[1 + i % 38], co[1 + i % 38])
torch.cuda.synchronize()
before = eng.telemetry[16:].view(-1, 16).clone()
T = 10
for i in range(T): eng.step(od[1 + (30 + i) % 38], co[1 + (30 + i) % 38])
torch.cuda.synchronize()
dd = (eng.telemetry[16:].view(-1, 16) - before).cpu().numpy().astype(float) / T
nw = dd.shape[0]
names = ["prop", "nn_solo", "nn_coop(+tree)", "mesh_solo", "mesh_coop", "tree3", "gather", "reduce"]
print("waves", nw, "mean wave lifetime us", dd[:, 7].mean() / 100.0)
print("ticks per wave per frame:", {n: int(v) for n, v in zip(names, dd[:, 8:16].mean(0))})
print("per wave per frame: nn coop lanes %.1f, mesh coop lanes %.1f, nn records scanned (solo) %.0f" % (dd[:, 2].mean(), dd[:, 3].mean(), dd[:, 6].mean()))
print("fallbacks per frame:", (eng.telemetry[:2].cpu().numpy() / 40.0).tolist())
