#!/usr/bin/env python3
"""The device replica of torch's CPU mt19937: a frame's 2 N words handed out (k_mt_blocks with stores + k_mt_emit) against the same
number of words stepped over (the block recurrence alone).  usage: tools/bench_mt_skip.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.torch_rng import TorchCpuStream
dev = torch.device("cuda", 0)
N = 100_000
g = TorchCpuStream(1234, device=dev, overlap=False)
for name in ("draw", "skip", "draw", "skip"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        if name == "draw":
            g.rand64(N)
        else:
            g.skip_words(2 * N); g.rand64(16)
    torch.cuda.synchronize()
    print(name, "%.1f us per frame's worth of words" % ((time.perf_counter() - t0) / 50 * 1e6))
