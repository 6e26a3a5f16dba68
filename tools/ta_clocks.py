#!/usr/bin/env python3
"""Phase clocks of one k_tail_a2d workgroup (block 12) on the bench workload, steady state (debug).
Needs a library built with -DMIDAS_DEBUG_CLOCKS:  tools/variants.sh dbg "-DMIDAS_DEBUG_CLOCKS" ;
MIDAS_HIP_LIB=$PWD/midastouch_amd/csrc/build/variants/dbg.so python tools/ta_clocks.py"""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import _lib
from midastouch_amd.engine import PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
N, K, D = 100_000, 50_000, 512
cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
T = 120
traj = make_trajectory(cb, T=T, seed=2001)
eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4000, device=dev)
rng = np.random.default_rng(100)
d0 = np.linalg.norm(cb.poses[:, :3, 3] - traj.gt_poses[0][:3, 3], axis=1)
near = np.argsort(d0)[: max(64, K // 20)]
eng.set_particles(torch.as_tensor(cb.poses[rng.choice(near, N)]))
eng.project_to_codebook()
odoms, codes = (torch.as_tensor(x).to(dev) for x in (traj.odoms, traj.codes))
lib = _lib.load()
lib.midas_debug_ta_clocks.argtypes = [ctypes.c_void_p]
out = (ctypes.c_longlong * 16)()
acc, wall, cnt = np.zeros(7), np.zeros(4), 0
for t in range(1, T):
    eng.step(odoms[t], codes[t])
    torch.cuda.synchronize()
    lib.midas_debug_ta_clocks(out)
    c = np.array(out[:8], dtype=np.float64)
    if t > 40:
        acc += np.diff(c); cnt += 1
        w = np.array(out[8:12], dtype=np.float64); wall += (w - w.min()) / 100.0
print("mean shader ticks of workgroup 12: [index loads issued, gather landed + extrema, barrier pair + block extrema, exponentials, e stored, block total + scan + tables, end]")
print((acc / cnt).round(0).tolist(), "sum", round(acc.sum() / cnt))
print("wall us: wg0 start, end, wg24 start, end:", (wall / cnt).round(2).tolist())
