"""Reference-SHAPED CPU implementation of one filter frame (TEST INFRASTRUCTURE / cpu_baseline only).

Does what the reference's loop body really does on a CPU (filter/filter.py:150-190), with the same
data movement: 6-d NN -> gather an (N, D) float64 matrix of codebook rows -> cosine over it ->
softmax -> mesh prune -> torch.multinomial resample.  It uses the same library calls as the
reference (torch CPU ops with all intra-op threads; torch.multinomial through WeightedRandomSampler's
code path) and stands in for the two third-party trees that are not installed here:
pynanoflann (n_jobs=16) -> scipy cKDTree.query(workers=-1); sklearn KDTree -> the same cKDTree class.
The SO(3) log-map (theseus) comes from the oracle's C restatement.

Only bench.py's cpu_baseline leg and tests/ may import this module.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.spatial import cKDTree

from . import oracle as orc


class RefShapedFilter:
    def __init__(self, cb_poses, cb_embeddings, mesh_vertices, sig_t=2e-4, sig_r=0.5, pen_max=0.002, workers=-1):
        self.poses = torch.as_tensor(cb_poses).float()
        self.embeddings = torch.as_tensor(cb_embeddings).double()  # reference storage: float64 (K, D)
        self.feat = orc.R3_SE3(np.asarray(cb_poses))
        self.tree = cKDTree(self.feat)
        self.mesh_tree = cKDTree(np.asarray(mesh_vertices, dtype=np.float64))
        self.sig_t, self.sig_r, self.pen_max, self.workers = sig_t, sig_r, pen_max, workers

    @staticmethod
    def _euler_zyx(rot_deg: torch.Tensor) -> torch.Tensor:
        a = torch.deg2rad(rot_deg)
        cz, sz = torch.cos(a[:, 0]), torch.sin(a[:, 0])
        cy, sy = torch.cos(a[:, 1]), torch.sin(a[:, 1])
        cx, sx = torch.cos(a[:, 2]), torch.sin(a[:, 2])
        one, zero = torch.ones_like(cz), torch.zeros_like(cz)
        Rz = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], -1).reshape(-1, 3, 3)
        Ry = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], -1).reshape(-1, 3, 3)
        Rx = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], -1).reshape(-1, 3, 3)
        return Rz @ Ry @ Rx

    def motion_model(self, poses: torch.Tensor, odom: torch.Tensor) -> torch.Tensor:
        n = poses.shape[0]
        tn = torch.normal(mean=0.0, std=self.sig_t, size=(n, 3))
        rot = torch.normal(mean=0.0, std=self.sig_r, size=(n, 3))
        Tn = torch.zeros((n, 4, 4))
        Tn[:, :3, :3], Tn[:, :3, 3], Tn[:, 3, 3] = self._euler_zyx(rot), tn, 1
        return poses @ (odom[None] @ Tn)

    def se3_nn(self, poses: torch.Tensor):
        q = orc.R3_SE3(poses.numpy())
        _, idx = self.tree.query(q, k=1, workers=self.workers)
        idx = torch.as_tensor(idx.astype(np.int64))
        return idx, self.embeddings[idx, :]  # the (N, D) float64 gather of tactile_tree.py:54-58

    def step(self, poses: torch.Tensor, odom: torch.Tensor, code: torch.Tensor):
        poses = self.motion_model(poses, odom)
        idx, nn_codes = self.se3_nn(poses)
        w = torch.nn.functional.cosine_similarity(torch.atleast_2d(code), nn_codes).squeeze()
        w = torch.nn.Softmax(dim=0)(w)
        dist, _ = self.mesh_tree.query(poses[:, :3, 3].numpy().astype(np.float64), k=1, workers=self.workers)
        m = torch.ones(len(w))
        m[torch.as_tensor(dist) > self.pen_max] = 0.0
        w = w * m
        p = w / torch.sum(w)
        if torch.all(p == 0) or torch.any(torch.isnan(p)):
            return poses, w
        ridx = torch.multinomial(p.double(), len(p), True)
        return poses[ridx], w[ridx]
