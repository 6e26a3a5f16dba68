#!/usr/bin/env python3
"""The device replica of torch's CPU generator alone: N = 100k float64 uniforms per call, sequential walk against pieces side by side
   (python tools/bench_mt_pieces.py [pieces ...]; under tools/prof_stats.sh for the per-kernel split)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.torch_rng import TorchCpuStream
dev = torch.device("cuda", 0)
N = 100_000
for pieces in [int(a) for a in sys.argv[1:]] or [0, 4, 8, 16]:
    st = TorchCpuStream(3000, dev, pieces=pieces)
    out = torch.empty(N, dtype=torch.float64, device=dev)
    for _ in range(3):
        st.rand64(N, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        st.rand64_async(N, out)
    torch.cuda.synchronize()
    print(f"pieces={pieces}: {(time.perf_counter() - t0) / 100 * 1e6:.1f} us per call of N={N}", flush=True)
