#!/usr/bin/env python3
"""us/step of the fused single-GPU step at a given particle count:  tools/bench_n.py N [K D]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.engine import FilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
N = int(sys.argv[1]); K = int(sys.argv[2]) if len(sys.argv) > 2 else 50000; D = int(sys.argv[3]) if len(sys.argv) > 3 else 512
dev = torch.device("cuda", 0)
cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001); tr = make_trajectory(cb, T=130, seed=2001)
eng = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
rng = np.random.default_rng(0)
d0 = np.linalg.norm(cb.poses[:, :3, 3] - tr.gt_poses[0][:3, 3], axis=1)
eng.set_particles(torch.as_tensor(cb.poses[rng.choice(np.argsort(d0)[: max(64, K // 20)], N)])); eng.project_to_codebook()
od, co = torch.as_tensor(tr.odoms).to(dev), torch.as_tensor(tr.codes).to(dev)
for i in range(10): eng.step(od[1 + i % 128], co[1 + i % 128])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100): eng.step(od[1 + (10 + i) % 128], co[1 + (10 + i) % 128])
torch.cuda.synchronize()
print(f"N={N} K={K} D={D}: {(time.perf_counter() - t0) / 100 * 1e6:.1f} us/step  TB2_TAB={os.environ.get('MIDAS_TB2_TAB', 'auto')} OVERLAP={os.environ.get('MIDAS_OVERLAP', '1')}")
