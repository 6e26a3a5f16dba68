#!/usr/bin/env python3
"""The codebook stream by itself (midas_score: k_score_reg) against the same stream inside the front kernel (MIDAS_DENSE_SCORES=1),
at K = 50k and K = 500k: what the fused form loses to its launch shape.  usage: tools/bench_dense_stream.py"""
import json, os, sys, time
os.environ["MIDAS_DENSE_SCORES"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import ops
from midastouch_amd.engine import PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
N, D = 100_000, 512
out = {}
for name, K, seed in (("004_sugar_box", 50_000, 1001), ("025_mug", 500_000, 1004)):
    cb = make_codebook(name, K=K, D=D, seed=seed); tr = make_trajectory(cb, T=70, seed=2004)
    C = ops.Codebook(torch.as_tensor(cb.embeddings).to(dev))
    code = torch.as_tensor(tr.codes[1]).to(dev)
    for _ in range(3): C.score(code)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): C.score(code)
    e1.record(); torch.cuda.synchronize()
    us_score = 1e3 * e0.elapsed_time(e1) / 20
    eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
    d0 = np.linalg.norm(cb.poses[:, :3, 3] - tr.gt_poses[0][:3, 3], axis=1)
    near = np.argsort(d0)[: K // 20]
    eng.set_particles(torch.as_tensor(cb.poses[np.random.default_rng(4).choice(near, N)])); eng.project_to_codebook()
    od, co = torch.as_tensor(tr.odoms).to(dev), torch.as_tensor(tr.codes).to(dev)
    for i in range(10): eng.step(od[1 + i % 68], co[1 + i % 68])
    torch.cuda.synchronize(); t3 = time.perf_counter()
    n = 50
    eng.run(od[11:11 + n], co[11:11 + n])
    torch.cuda.synchronize()
    us = (time.perf_counter() - t3) / n * 1e6
    stream = K * (4 * D + 8 + 8)
    out[f"K{K // 1000}k"] = {"score_alone_us": round(us_score, 1), "score_alone_TBps": round(stream / us_score / 1e6, 2),
                             "dense_frame_us": round(us, 1), "frame_TBps_on_stream_bytes": round(stream / us / 1e6, 2)}
    del eng, C
print(json.dumps(out))
