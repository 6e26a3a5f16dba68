#!/usr/bin/env python3
"""After the wide start of bench.py: of the codebook rows a frame needs that are NOT on its prediction list (rows used in the four
frames before), how many are among the first M records of the neighbour list of a row that WAS used in the frame before?  (Would
listing the neighbours of the rows in use take the claims off the particle waves?  And what would the lists grow to?)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.engine import PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory, wide_start
from oracle import oracle as orc
dev = torch.device("cuda", 0)
N, K, D = 100_000, 50_000, 512
cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
traj = make_trajectory(cb, T=140, seed=2001)
eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4000, device=dev)
eng.set_particles(torch.as_tensor(wide_start(cb.extents, traj.gt_poses[0], N, 100)))
eng.project_to_codebook()
odoms, codes, gts = (torch.as_tensor(x).to(dev) for x in (traj.odoms, traj.codes, traj.gt_poses))
# the first 16 neighbours of every entry (k-NN in the 6-d feature space, as the lists hold them)
feat = orc.R3_SE3(cb.poses)
from midastouch_amd import ops
nb = ops.knn6(eng.tree6, torch.as_tensor(feat).to(dev), 17).cpu().numpy()[:, 1:]  # (K, 16) without the entry itself
hist = []
print("frame rows_in_use not_listed  covered_by_M=2 M=4 M=8 M=16   list_size_with_M=4 M=8")
for t in range(1, 32):
    eng.step(odoms[t], codes[t], gt=gts[t])
    eng.flush()
    rows = np.unique(eng._nn[eng._cur].cpu().numpy()[:N])
    if len(hist) >= 4:
        listed = np.unique(np.concatenate(hist[-4:]))
        new = np.setdiff1d(rows, listed)
        prev = hist[-1]
        cov = []
        sizes = []
        for M in (2, 4, 8, 16):
            cand = np.unique(nb[prev][:, :M].ravel())
            cov.append(int(np.isin(new, cand).sum()))
            if M in (4, 8):
                sizes.append(len(np.union1d(listed, cand)))
        print("%4d %8d %8d      %6d %6d %6d %6d      %8d %8d" % (t, len(rows), len(new), *cov, *sizes), "listed", len(listed))
    hist.append(rows)
