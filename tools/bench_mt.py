#!/usr/bin/env python3
"""Duration of the device replica of torch's CPU generator: N float64 uniforms (and a skip of the two torch.normal calls of a frame)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.torch_rng import TorchCpuStream, normal_words
dev = torch.device("cuda", 0)
st = TorchCpuStream(3000, dev)
for N in (10_000, 100_000, 1_000_000):
    out = torch.empty(N, dtype=torch.float64, device=dev)
    for skipn in (0, 2 * normal_words(3 * N)):
        st.rand64(N, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            st.skip_words(skipn).rand64(N, out)
        e1.record()
        torch.cuda.synchronize()
        us = 100.0 * e0.elapsed_time(e1)
        print(f"N={N} skip={skipn}: {us:.1f} us per call, {us * 1e3 / ((2 * N + skipn) / 624):.0f} ns per 624-word block")
