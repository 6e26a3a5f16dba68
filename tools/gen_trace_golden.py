#!/usr/bin/env python3
"""G10: a T-frame trace of the reference's loop body (filter/filter.py:150-190) driven through the REAL reference
functions (add_noise_to_odom + compose, get_similarity, remove_invalid_particles, resampler) with the two
operations the reference cannot run here (theseus SO3 log-map, pynanoflann KD-tree inside SE3_NN) supplied by
the oracle.  Runs only in the build container (imports /root/reference); writes tests/golden/g10_trace.npz.

Per frame it stores the particle poses before and after the motion model, the NN indices, the weights after
get_similarity and after the prune, and the resample indices, so that a test can teacher-force every stage.
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
from gen_goldens import import_reference, new_pf  # noqa: E402

from midastouch_amd.synthetic import make_codebook, make_trajectory, mesh_scale  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    from sklearn.neighbors import KDTree
    torch.set_num_threads(1)
    pfm, _ = import_reference()
    N, K, D, T = 256, 1200, 256, 24
    cb = make_codebook(K=K, D=D, seed=1000, mesh_points=20000)
    traj = make_trajectory(cb, T=T + 1, seed=2000)
    pf = new_pf(pfm)
    pf.mesh_kdtree = KDTree(cb.mesh_vertices)
    cb_feat = orc.R3_SE3(cb.poses)
    emb64 = torch.tensor(cb.embeddings).double()
    # t = 0: init_filter arithmetic (scale of the synthetic box) then projection onto the codebook
    pf.init_noise = [mesh_scale(cb.extents) / 3.0 * 0.05, 60.0 * 0.05]
    torch.manual_seed(100)
    parts = pf.init_filter(torch.tensor(traj.gt_poses[0]), N)
    idx0 = orc.nn6(orc.R3_SE3(parts.poses.numpy()), cb_feat)[0]
    parts.poses = torch.tensor(cb.poses[idx0])
    out = {"N": N, "K": K, "D": D, "T": T, "cb_seed": 1000, "traj_seed": 2000, "poses0": parts.poses.numpy().copy()}
    for t in range(1, T + 1):
        seed = 3000 + t
        torch.manual_seed(seed)
        odom = torch.tensor(traj.odoms[t])
        rep = torch.repeat_interleave(odom[None], N, dim=0)
        noisy = pf.add_noise_to_odom(rep, mul=1.0)           # reference: draws tn, rot from the CPU generator
        prop = parts.poses @ noisy                            # reference compose (motionModel :374)
        parts = pfm.Particles(prop, parts.weights, parts.labels)
        nn_idx = orc.nn6(orc.R3_SE3(prop.numpy()), cb_feat)[0]  # oracle stands in for SE3_NN's tree
        code = torch.tensor(traj.codes[t])[None]
        w_sim = pf.get_similarity(code, emb64[torch.as_tensor(nn_idx.astype(np.int64))], softmax=True)
        parts.weights = w_sim.clone()
        parts, drifted = pf.remove_invalid_particles(parts)
        w_pruned = parts.weights.clone()
        marker = pfm.Particles(parts.poses.clone(), parts.weights, torch.arange(N, dtype=torch.float32))
        res = pf.resampler(marker)                            # consumes torch.multinomial's draws
        ridx = res.labels.numpy().astype(np.int32)
        out[f"pre_{t}"] = parts.poses.numpy().copy() if False else None
        out[f"prop_{t}"] = prop.numpy().copy()
        out[f"nn_{t}"] = nn_idx.astype(np.int32)
        out[f"wsim_{t}"] = w_sim.numpy().copy()
        out[f"wprune_{t}"] = w_pruned.numpy().copy()
        out[f"ridx_{t}"] = ridx
        out[f"drifted_{t}"] = np.bool_(bool(drifted))
        rt, rr = pfm.particle_rmse(pfm.Particles(prop), torch.tensor(traj.gt_poses[t]))
        out[f"rmse_{t}"] = np.array([rt.item(), rr.item()], dtype=np.float32)
        parts = pfm.Particles(res.poses, res.weights, torch.zeros(N))
    out = {k: v for k, v in out.items() if v is not None}
    path = os.path.join(REPO, "tests", "golden", "g10_trace.npz")
    np.savez_compressed(path, **out)
    print("g10_trace: %.1f KiB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
