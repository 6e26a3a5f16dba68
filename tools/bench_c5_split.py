#!/usr/bin/env python3
"""c5 as G independent sub-batches of B / G trajectories, each on a HIP stream (and library context) of its own: one group's
latency-bound kernels (presort, tail) run beside another group's particle waves.  GPU box only."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import _lib
from midastouch_amd.engine import PipelinedBatchFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
cb = make_codebook("cotter-pin", K=50000, D=512, seed=1005)
B, N = 64, 10000
trs = [make_trajectory(cb, T=40, seed=2200 + b) for b in range(8)]
od = torch.as_tensor(np.stack([trs[b % 8].odoms for b in range(B)], axis=1)).to(dev)
co = torch.as_tensor(np.stack([trs[b % 8].codes for b in range(B)], axis=1)).to(dev)
for G in (1, 2, 4):
    Bg = B // G
    streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
    engs = []
    rng = np.random.default_rng(1)
    start = np.stack([cb.poses[rng.integers(0, 50000, N)] for _ in range(B)])
    for g in range(G):
        e = PipelinedBatchFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, Bg, N, seed=4000 + g, device=dev)
        e.set_particles(torch.as_tensor(start[g * Bg:(g + 1) * Bg])); e.project_to_codebook()
        engs.append(e)
    torch.cuda.synchronize()
    for g in range(G):
        with torch.cuda.stream(streams[g]):
            engs[g].ctx = _lib.Context(dev)  # a context (scratch, stream binding) of its own
    ods = [od[:, g * Bg:(g + 1) * Bg].contiguous() for g in range(G)]
    cos = [co[:, g * Bg:(g + 1) * Bg].contiguous() for g in range(G)]
    torch.cuda.synchronize()
    def steps(i0, n):
        for i in range(i0, i0 + n):
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    engs[g].step(ods[g][1 + i % 38], cos[g][1 + i % 38])
    steps(0, 10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    steps(10, 60)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 60 * 1e6
    print(f"c5 in {G} group(s) of {Bg}: {us:.1f} us per batch step of {B}, {B * 1e6 / us:.0f} trajectory-steps/s")
