#!/usr/bin/env python3
"""c5 (B=64 x N=10k, K=50k, D=512) batch-step time with particles spread over the object or started near the truth."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.engine import BatchFilterEngine, PipelinedBatchFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
cb = make_codebook("cotter-pin", K=50000, D=512, seed=1005)
B, N = 64, 10000
trs = [make_trajectory(cb, T=40, seed=2200 + b) for b in range(8)]
od = torch.as_tensor(np.stack([trs[b % 8].odoms for b in range(B)], axis=1)).to(dev)
co = torch.as_tensor(np.stack([trs[b % 8].codes for b in range(B)], axis=1)).to(dev)
ENG = PipelinedBatchFilterEngine if os.environ.get("MIDAS_C5_PIPELINED", "1") != "0" else BatchFilterEngine
for init in ("spread", "near"):
    eng = ENG(cb.poses, cb.embeddings, cb.mesh_vertices, B, N, device=dev)
    rng = np.random.default_rng(1)
    if init == "spread":
        start = np.stack([cb.poses[rng.integers(0, 50000, N)] for _ in range(B)])
    else:
        start = []
        for b in range(B):
            d0 = np.linalg.norm(cb.poses[:, :3, 3] - trs[b % 8].gt_poses[0][:3, 3], axis=1)
            start.append(cb.poses[rng.choice(np.argsort(d0)[:2500], N)])
        start = np.stack(start)
    eng.set_particles(torch.as_tensor(start)); eng.project_to_codebook()
    for i in range(10): eng.step(od[1 + i % 38], co[1 + i % 38])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(60): eng.step(od[1 + (10 + i) % 38], co[1 + (10 + i) % 38])
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 60 * 1e6
    tele = eng.telemetry.cpu().numpy()[:2] / 70.0
    print(f"   tree fallbacks per batch step: nn {tele[0]:.1f}, prune {tele[1]:.1f}")
    print(f"c5 init={init}: {us:.1f} us per batch step, {B * 1e6 / us:.0f} trajectory-steps/s, engine={ENG.__name__}")
