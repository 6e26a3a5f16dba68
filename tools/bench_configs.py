#!/usr/bin/env python3
"""Timing of the non-headline BASELINE configs on one GPU (GPU box only): c1 (N=1k,K=5k,D=256), c5 batch
(B=64 x N=10k, K=50k, D=512) and a 1M-particle single-GPU frame (c3's total on one device)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.engine import BatchFilterEngine, FilterEngine, PipelinedBatchFilterEngine, PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)


def run(eng, step, n=100, warm=10):
    for i in range(warm): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): step(warm + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


res = {}
cb = make_codebook(K=5000, D=256, seed=1000); tr = make_trajectory(cb, T=130, seed=2000)
od, co = torch.as_tensor(tr.odoms).to(dev), torch.as_tensor(tr.codes).to(dev)
for tag, cls in (("eager", FilterEngine), ("pipelined", PipelinedFilterEngine)):
    eng = cls(cb.poses, cb.embeddings, cb.mesh_vertices, 1000, device=dev)
    eng.set_particles(torch.as_tensor(cb.poses[np.random.default_rng(0).integers(0, 5000, 1000)])); eng.project_to_codebook()
    us = run(eng, lambda i: eng.step(od[1 + i % 128], co[1 + i % 128]))
    res["c1_N1k_K5k_D256_" + tag] = {"us_per_step": round(us, 1), "steps_per_s": round(1e6 / us)}

cb = make_codebook("cotter-pin", K=50000, D=512, seed=1005)
B, N = 64, 10000
trs = [make_trajectory(cb, T=40, seed=2200 + b) for b in range(8)]
od = torch.as_tensor(np.stack([trs[b % 8].odoms for b in range(B)], axis=1)).to(dev)   # (T,B,4,4)
co = torch.as_tensor(np.stack([trs[b % 8].codes for b in range(B)], axis=1)).to(dev)
for tag, cls in (("", PipelinedBatchFilterEngine), ("_eager", BatchFilterEngine)):
    eng = cls(cb.poses, cb.embeddings, cb.mesh_vertices, B, N, device=dev)
    rng = np.random.default_rng(1)
    eng.set_particles(torch.as_tensor(np.stack([cb.poses[rng.integers(0, 50000, N)] for _ in range(B)]))); eng.project_to_codebook()
    us = run(eng, lambda i: eng.step(od[1 + i % 38], co[1 + i % 38]), n=60)
    res["c5_B64_N10k_K50k_D512" + tag] = {"us_per_batch_step": round(us, 1), "trajectory_steps_per_s": round(B * 1e6 / us)}
    del eng

cb = make_codebook("035_power_drill", K=50000, D=512, seed=1003); tr = make_trajectory(cb, T=70, seed=2003)
N = 1_000_000
od, co = torch.as_tensor(tr.odoms).to(dev), torch.as_tensor(tr.codes).to(dev)
for tag, cls in (("eager", FilterEngine), ("pipelined", PipelinedFilterEngine)):
    eng = cls(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
    eng.set_particles(torch.as_tensor(cb.poses[np.random.default_rng(2).integers(0, 50000, N)])); eng.project_to_codebook()
    us = run(eng, lambda i: eng.step(od[1 + i % 68], co[1 + i % 68]), n=40)
    res["c3total_N1M_K50k_D512_single_gpu_" + tag] = {"us_per_step": round(us, 1), "steps_per_s": round(1e6 / us)}
    del eng
print(json.dumps(res))
