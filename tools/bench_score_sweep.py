#!/usr/bin/env python3
"""GEMV scoring rate vs codebook size (GPU box only): is a 102 MB codebook served from HBM or from the 256 MB Infinity Cache?"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import ops
dev = torch.device("cuda", 0)
D = 512
rng = np.random.default_rng(0)
res = {}
for K in (6_250, 12_500, 25_000, 50_000, 100_000, 200_000, 400_000):
    E = torch.randn((K, D), device=dev)
    cb = ops.Codebook(E)
    code = torch.randn((1, D), dtype=torch.float64, device=dev)
    for _ in range(5): cb.score(code)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    s.record()
    for _ in range(n): cb.score(code)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / n * 1e3
    res[f"K{K}"] = {"MB": round(K * D * 4 / 1e6, 1), "us": round(us, 2), "GBps": round(K * D * 4 / us / 1e3)}
    del cb, E
print(json.dumps(res))
