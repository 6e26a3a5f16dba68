#!/usr/bin/env python3
"""Batched scoring (BASELINE config 5 shape: B codes x K x D): MFMA kernel vs the GEMV loop (GPU box only)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import ops
dev = torch.device("cuda", 0)
K, D = 50_000, 512
rng = np.random.default_rng(0)
E = rng.standard_normal((K, D)).astype(np.float32); E /= np.linalg.norm(E, axis=1, keepdims=True)
cb = ops.Codebook(torch.as_tensor(E).to(dev))
res = {}
for B in (1, 16, 64):
    codes = torch.as_tensor(rng.standard_normal((B, D)).astype(np.float32)).double().to(dev)
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    t_m, t_g = timeit(lambda: cb.score_batch(codes)), timeit(lambda: cb.score(codes))
    flop = 2.0 * K * D * B
    res[f"B{B}"] = {"mfma_us": round(t_m, 1), "gemv_loop_us": round(t_g, 1), "mfma_TFLOPs": round(flop / t_m / 1e6, 1),
                    "mfma_GBps": round(K * D * 4 / t_m / 1e3, 1)}
print(json.dumps(res))
