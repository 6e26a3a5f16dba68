#!/usr/bin/env python3
"""G10b / G13: T = 64 frame traces of the reference's loop body (filter/filter.py:150-190) at N = 4096, driven through
the REAL reference functions (build container only; imports /root/reference, see tools/gen_goldens.py).

  G10b  fixed particle count: get_similarity, remove_invalid_particles, resampler, particle_rmse.
  G13   the whole body: + cluster_particles (DBSCAN every 50th frame), annealing - the particle count changes from
        frame to frame.  get_cluster_centers("quat_avg") cannot run here (removed Tensor.eig, theseus): the cluster
        centres come from the oracle (pinned against scipy), `torch.mean(cluster_stds)` and everything downstream of
        it from the reference.

The motion model of these two traces is the oracle's fixed-order float32 compose fed with the reference's draw order
(torch.normal tn, rot on the CPU generator) - the reference's own compose is pinned by G3 and the T = 24 trace G10 - so
that every frame's particle set is exactly reproducible from the previous one and the fixture can hold digests
(SHA-256 + head / tail) instead of the arrays.  Per frame: seed 3000 + t, draws tn, rot, then the
resampler's N' float64 uniforms, as in G10.

Ties: torch.topk leaves the choice among equal weights at the k-th position to the implementation; the spec here
breaks them by index.  A frame where the reference's choice differs only inside such a tie is flagged (`tie_t`) and its
kept-index list stored, so that a replay can follow the reference's choice.
"""
import copy
import hashlib
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

K, D, T, N0 = 3000, 256, 64, 4096
CB_SEED, TRAJ_SEED = 1013, 2013


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def digest(out, key, a):
    a = np.ascontiguousarray(a)
    out[key + "_sha"] = np.str_(sha(a))
    out[key + "_head"], out[key + "_tail"] = a[:32].copy(), a[-32:].copy()


def run(pfm, cluster: bool, name: str, init_ratio: float):
    from sklearn.neighbors import KDTree
    cb = make_codebook(K=K, D=D, seed=CB_SEED, mesh_points=20000)
    traj = make_trajectory(cb, T=T + 1, seed=TRAJ_SEED)
    pf = new_pf(pfm)
    pf.mesh_kdtree = KDTree(cb.mesh_vertices)
    shadow = new_pf(pfm)  # the same annealing on index markers: which particles the reference kept
    cb_feat = orc.R3_SE3(cb.poses)
    emb64 = torch.tensor(cb.embeddings).double()
    pf.init_noise = [mesh_scale(cb.extents) / 3.0 * init_ratio, 60.0 * init_ratio]
    torch.manual_seed(100)
    parts = pf.init_filter(torch.tensor(traj.gt_poses[0]), N0)
    idx0 = orc.nn6(orc.R3_SE3(parts.poses.numpy()), cb_feat)[0]
    poses = cb.poses[idx0].copy()
    labels = np.zeros(N0, dtype=np.int64)
    out = {"N0": N0, "K": K, "D": D, "T": T, "cb_seed": CB_SEED, "traj_seed": TRAJ_SEED, "poses0": poses.copy(),
           "cluster": np.bool_(cluster), "cb_sha": np.str_(sha(cb.embeddings.astype(np.float32)))}
    oracle_ann = orc.Annealer()
    aten_ann = orc.Annealer(ties="aten_cpu")  # the restatement of ATen's CPU top-k: must make the reference's choice in EVERY frame
    ties = 0
    for t in range(1, T + 1):
        N = poses.shape[0]
        torch.manual_seed(3000 + t)
        tn = torch.normal(mean=0.0, std=2e-4, size=(N, 3)).numpy()      # add_noise_to_odom's draws, its order (:326-335)
        rot = torch.normal(mean=0.0, std=0.5, size=(N, 3)).numpy()
        prop = orc.propagate(poses, traj.odoms[t], tn, rot)
        rt, rr = pfm.particle_rmse(pfm.Particles(torch.tensor(prop)), torch.tensor(traj.gt_poses[t]))
        nn_idx = orc.nn6(orc.R3_SE3(prop), cb_feat)[0]
        code = torch.tensor(traj.codes[t])[None]
        w_sim = pf.get_similarity(code, emb64[torch.as_tensor(nn_idx.astype(np.int64))], softmax=True)
        parts = pfm.Particles(torch.tensor(prop), w_sim.clone(), torch.tensor(labels))
        parts, drifted = pf.remove_invalid_particles(parts)
        if bool(drifted):  # filter.py:176-179
            prop = cb.poses[nn_idx].copy()
            parts.poses = torch.tensor(prop)
        w_pruned = parts.weights.clone().numpy()
        out[f"N_{t}"] = np.int64(N)
        out[f"rmse_{t}"] = np.array([rt.item(), rr.item()], dtype=np.float32)
        digest(out, f"nn_{t}", nn_idx.astype(np.int32))
        digest(out, f"wsim_{t}", w_sim.numpy())
        digest(out, f"wprune_{t}", w_pruned)
        out[f"drifted_{t}"] = np.bool_(bool(drifted))
        keep = np.arange(N)
        if cluster:
            if (t - 1) % 50 == 0:  # count % 50 == 0, count = 0 on the first frame
                parts = pf.cluster_particles(parts)
                labels = parts.labels.numpy().astype(np.int64)
                digest(out, f"dbscan_{t}", labels.astype(np.int32))
            uniq, centers, stds = orc.cluster_centers(prop, w_pruned, labels)
            var = torch.mean(torch.tensor(stds))                       # filter.py:189
            assert np.float32(var.item()) == orc.cluster_var(stds), "torch.mean differs from the spec's float32 sum"
            out[f"cl_labels_{t}"], out[f"cl_poses_{t}"], out[f"cl_stds_{t}"] = uniq.astype(np.int32), centers, stds
            out[f"var_{t}"] = np.float32(var.item())
            shadow.particle_var = copy.copy(pf.particle_var)
            if hasattr(pf, "init_particles"):
                shadow.init_particles = pf.init_particles
            marker = pfm.Particles(torch.tensor(prop), parts.weights.clone(), torch.arange(N, dtype=torch.float64))
            parts = pf.annealing(parts, var)
            keep = shadow.annealing(marker, var).labels.numpy().astype(np.int64)
            assert len(keep) == len(parts) and torch.equal(parts.poses, torch.tensor(prop)[keep])
            spec_keep = oracle_ann.step(w_pruned, np.float32(var.item()))
            assert np.array_equal(aten_ann.step(w_pruned, np.float32(var.item())), keep), f"frame {t}: aten_topk restatement differs from torch.topk"
            tie = not np.array_equal(spec_keep, keep)
            if tie:  # the two choices must differ only among equal weights
                assert len(spec_keep) == len(keep) and np.array_equal(np.sort(w_pruned[spec_keep]), np.sort(w_pruned[keep])), t
                out[f"keep_{t}"] = keep.astype(np.int32)
                ties += 1
            out[f"tie_{t}"] = np.bool_(tie)
            digest(out, f"keep_{t}", keep.astype(np.int32))
        n2 = len(parts)
        carried = parts.labels.clone()
        parts.labels = torch.arange(n2, dtype=torch.float64)             # marker: which slot each draw took
        res = pf.resampler(parts)                                        # consumes torch.multinomial's n2 draws
        ridx = res.labels.numpy().astype(np.int32)
        digest(out, f"ridx_{t}", ridx)
        out[f"N2_{t}"] = np.int64(n2)
        src = keep[ridx]
        poses = prop[src]
        assert np.array_equal(res.poses.numpy(), poses)
        labels = carried.numpy().astype(np.int64)[ridx]
        print(f"  {name} t={t:2d} N={N:5d} -> {n2:5d} rmse_t={1e3 * rt.item():6.2f} mm kept={(w_pruned > 0).sum():5d}"
              + (f" clusters={list(uniq)} var={var.item():.3e} tie={tie}" if cluster else ""))
    path = os.path.join(REPO, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, tie frames {ties}")


def main():
    torch.set_num_threads(1)
    pfm, _ = import_reference()
    run(pfm, False, "g10b_trace64", 0.05)
    run(pfm, True, "g13_loop_trace", 0.15)


if __name__ == "__main__":
    main()
