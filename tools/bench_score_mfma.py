#!/usr/bin/env python3
"""midas_score_batch (k_score_mfma) at c5's shape and a few others: us per call, TFLOP/s, GB/s of the codebook stream (GPU box only)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import ops
dev = torch.device("cuda", 0)
res = {}
for K, D, B in ((50_000, 512, 64), (50_000, 256, 64), (500_000, 512, 64), (50_000, 512, 16), (5_000, 256, 64), (50_000, 512, 128)):
    E = torch.randn((K, D), device=dev)
    cb = ops.Codebook(E)
    codes = torch.randn((B, D), dtype=torch.float64, device=dev)
    for _ in range(5): cb.score_batch(codes)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    s.record()
    for _ in range(n): cb.score_batch(codes)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / n * 1e3
    res[f"K{K}_D{D}_B{B}"] = {"us_per_call_incl_codes_prepare": round(us, 2), "TFLOPs": round(2.0 * K * D * B / us / 1e6, 1),
                              "codebook_GBps": round(K * D * 4 / us / 1e3)}
    del cb, E
print(json.dumps(res))
