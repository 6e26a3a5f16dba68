#!/usr/bin/env python3
"""top_n_error (eval/single_touch_test.py:35-73) at K = 50 000, D = 256, n = 25 - the reference's only dense GEMM (K x K x D =
1.28 TFLOP) - end to end, with the matrix-core rate of k_selfsim_mfma inside it (HIP events around a panel's GEMM alone).
usage: tools/bench_topn.py [K] [D] [panel_rows]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import ops
from midastouch_amd.single_touch import top_n_error
from midastouch_amd.synthetic import make_codebook
dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
R = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
emb = torch.as_tensor(cb.embeddings).to(dev)
poses = torch.as_tensor(cb.poses[:, :3, 3]).to(dev)
top_n_error(emb[:4096].contiguous(), poses[:4096].contiguous(), fast=True)  # warm-up (library, scratch)
torch.cuda.synchronize()
res = {}
for rows in (R, 4096):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    err = top_n_error(emb, poses, fast=True, panel_rows=rows)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[f"panel_{rows}"] = {"seconds": dt, "gemm_tflop": 2.0 * K * K * D / 1e12, "end_to_end_tflops": 2.0 * K * K * D / dt / 1e12, "mean_err_mm": float(err.mean()) * 1e3}
# the GEMM alone: one panel, HIP events
codebook = ops.Codebook(emb)
ldo = -(-K // 128) * 128
panel = torch.empty((-(-R // 128) * 128, ldo), dtype=torch.float32, device=dev)
lib, ctx = codebook.ctx.lib, codebook.ctx
import ctypes as C
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for R in (R, 4096, 8192):
    panel = torch.empty((-(-R // 128) * 128, ldo), dtype=torch.float32, device=dev)
    for _ in range(2):
        ctx.call("midas_selfsim_panel", codebook.h, 0, R, ops._ptr(panel), ldo)
    e0.record()
    for _ in range(5):
        ctx.call("midas_selfsim_panel", codebook.h, 0, R, ops._ptr(panel), ldo)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    res[f"k_selfsim_mfma_{R}"] = {"panel_rows": R, "ms": ms, "tflops": 2.0 * R * K * D / (ms * 1e-3) / 1e12, "frac_of_157.3": 2.0 * R * K * D / (ms * 1e-3) / 1e12 / 157.3}
print(json.dumps(res))
