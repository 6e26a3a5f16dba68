#!/usr/bin/env python3
"""G11: a small `codebook.pkl` written the way the reference writes it (tactile_tree/build_codebook.py:130-137:
`dill.dump(tactile_tree(poses=..., cam_poses=..., embeddings=...))`) + the arrays that went in, for the
converter test (tests/test_codebook_io.py).

Runs only in the build container: it imports the REAL `midastouch.tactile_tree.tactile_tree.tactile_tree` class from
/root/reference.  Two third-party pieces that class needs are not installed here and are stood in for - neither
decides anything the converter reads:
  * `pynanoflann.KDTree` -> a picklable object that keeps the fitted array (the converter drops the tree),
  * theseus `SO3.log_map` (behind `pose.get_logmap_from_matrix`) -> scipy `Rotation.as_rotvec` (the stored
    `logmap_pose` is therefore scipy's, float32).
The pickle holds names and tensors only (the class is pickled by reference), no reference source.
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "tests", "golden")


def main():
    import dill

    nf = types.ModuleType("pynanoflann")

    class KDTree:  # stand-in, see the module docstring
        def __init__(self, metric="L2", radius=1.0):
            self.metric, self.radius, self.data = metric, float(radius), None

        def fit(self, X):
            self.data = np.array(X)

    KDTree.__module__, KDTree.__qualname__ = "pynanoflann", "KDTree"  # pickled by reference, like the real class
    nf.KDTree = KDTree
    sys.modules["pynanoflann"] = nf
    for name in ["trimesh", "theseus"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, "/root/reference")
    from scipy.spatial.transform import Rotation

    import midastouch.tactile_tree.tactile_tree as tt

    tt.get_logmap_from_matrix = lambda R: torch.as_tensor(Rotation.from_matrix(R.numpy().astype(np.float64)).as_rotvec(),
                                                          dtype=torch.float32)
    from midastouch_amd.synthetic import make_codebook

    cb = make_codebook(K=300, D=64, seed=1100, mesh_points=2000)
    rng = np.random.default_rng(7)
    cam = cb.poses.copy()
    cam[:, :3, 3] += (0.022 * cam[:, :3, 2]).astype(np.float32)  # camera behind the gel along its z axis
    emb64 = torch.tensor(cb.embeddings).double()  # the reference's contract: float32 codes cast to float64
    tree = tt.tactile_tree(poses=torch.tensor(cb.poses), cam_poses=torch.tensor(cam), embeddings=emb64)
    path = os.path.join(OUT, "g11_codebook_ref.pkl")
    with open(path, "wb") as f:
        dill.dump(tree, f)
    # second object: embeddings that are NOT float32-representable (the container must keep float64)
    emb_odd = emb64 + torch.tensor(rng.standard_normal(emb64.shape) * 1e-12)
    tree2 = tt.tactile_tree(poses=torch.tensor(cb.poses[:40]), cam_poses=torch.tensor(cam[:40]), embeddings=emb_odd[:40].clone())
    with open(os.path.join(OUT, "g11_codebook_ref_f64.pkl"), "wb") as f:
        dill.dump(tree2, f)
    np.savez_compressed(os.path.join(OUT, "g11_codebook_arrays.npz"), poses=cb.poses, cam_poses=cam,
                        embeddings=cb.embeddings, embeddings_f64=emb_odd[:40].numpy(),
                        logmap_pose=tree.logmap_pose.numpy())
    for n in ("g11_codebook_ref.pkl", "g11_codebook_ref_f64.pkl", "g11_codebook_arrays.npz"):
        print(n, os.path.getsize(os.path.join(OUT, n)) // 1024, "KiB")


if __name__ == "__main__":
    main()
