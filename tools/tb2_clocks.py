#!/usr/bin/env python3
"""Phase clocks of one k_tail_b2 workgroup on the bench workload (debug).
Needs a library built with -DMIDAS_DEBUG_CLOCKS:  tools/variants.sh dbg "-DMIDAS_DEBUG_CLOCKS" ;
MIDAS_HIP_LIB=$PWD/midastouch_amd/csrc/build/variants/dbg.so python tools/tb2_clocks.py"""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import _lib
from midastouch_amd.engine import FilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
N, K, D = 100_000, 50_000, 512
cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
T = 60
traj = make_trajectory(cb, T=T, seed=2001)
eng = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4000, device=dev)
rng = np.random.default_rng(100)
d0 = np.linalg.norm(cb.poses[:, :3, 3] - traj.gt_poses[0][:3, 3], axis=1)
near = np.argsort(d0)[: max(64, K // 20)]
eng.set_particles(torch.as_tensor(cb.poses[rng.choice(near, N)]))
eng.project_to_codebook()
odoms, codes = (torch.as_tensor(x).to(dev) for x in (traj.odoms, traj.codes))
lib = _lib.load()
out = (ctypes.c_longlong * 16)()
acc = np.zeros(7)
for t in range(1, T):
    eng.step(odoms[t], codes[t])
    torch.cuda.synchronize()
    lib.midas_debug_tb2_clocks(out)
    c = np.array(out[:8], dtype=np.float64)
    if t > 10: acc += np.diff(c)
    w = np.array(out[8:14], dtype=np.float64); w = (w - w.min()) / 100.0
    if t > T - 4: print('wall us [wg0 start,end, wg195 start,end, wg390 start,end]', w.round(2).tolist())
print("mean ticks [loads issued, loads landed+guard+prefix, status, lds search, chunk fetch+fix-up test, walk, gathers]")
print((acc / (T - 11)).round(0).tolist())
