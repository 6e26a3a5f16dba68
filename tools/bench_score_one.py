#!/usr/bin/env python3
"""midas_score_batch at c5's shape only (K = 50k, D = 512, B = 64): the command the PMC passes of k_score_mfma profile (GPU box only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import ops
dev = torch.device("cuda", 0)
K, D, B = 50_000, 512, 64
cb = ops.Codebook(torch.randn((K, D), device=dev))
codes = torch.randn((B, D), dtype=torch.float64, device=dev)
for _ in range(5): cb.score_batch(codes)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(30): cb.score_batch(codes)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 30 * 1e3
print(f"K={K} D={D} B={B}: {us:.1f} us per call (incl. k_codes_prepare), {2.0 * K * D * B / us / 1e6:.1f} TFLOP/s")
