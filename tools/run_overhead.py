#!/usr/bin/env python3
"""Fixed cost of a timed region of T frames (what bench.py --steps 20 brackets): wall time between two synchronisations
against the device's own span (clock of the first and last frame in the run log) and the host time of the run() call."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.engine import PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
N, K, D = 100000, 50000, 512
cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
traj = make_trajectory(cb, T=400, seed=2001)
eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4000, device=dev)
rng = np.random.default_rng(0)
eng.set_particles(torch.as_tensor(cb.poses[rng.integers(0, K, N)]))
od, co, gt = (torch.as_tensor(x).to(dev) for x in (traj.odoms, traj.codes, traj.gt_poses))
eng.run(od[1:101], co[1:101], gt[1:101])
torch.cuda.synchronize()
for T in (20, 200):
    rows = []
    for rep in range(8):
        t0_ = 101 + (rep * T) % 150
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        log = eng.run(od[t0_:t0_ + T], co[t0_:t0_ + T], gt[t0_:t0_ + T])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts = log[:, 2].cpu().numpy()
        rows.append((1e3 * (t2 - t0), 1e3 * (t1 - t0), (ts[-1] - ts[0]) * 1e-3 * T / (T - 1)))
    r = np.array(rows)[2:]
    print(f"T={T}: wall {r[:,0].mean():.3f} ms, host call {r[:,1].mean():.3f} ms, device frames {r[:,2].mean():.3f} ms, "
          f"fixed cost {1e3 * (r[:,0] - r[:,2]).mean():.0f} us, per step wall {1e3 * r[:,0].mean() / T:.1f} us")
