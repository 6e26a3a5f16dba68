"""Tensor-level wrappers over the C ABI (one function per kernel group).

Inputs/outputs are torch tensors on a HIP device; torch only owns the memory.  Each function
cites the reference arithmetic it replaces (paths relative to /root/reference/midastouch).
"""
from __future__ import annotations

import ctypes as C
import sys

import torch

from . import _lib
from ._lib import MidasError, _ptr


def _ctx(t: torch.Tensor):
    if not t.is_cuda:
        raise MidasError("midastouch_amd kernels need tensors on a HIP device; there is no CPU fallback")
    return _lib.context(t.device)


def _poses(p: torch.Tensor) -> torch.Tensor:
    p = p[None] if p.dim() == 2 else p
    if p.dtype != torch.float32 or not p.is_contiguous():
        p = p.float().contiguous()
    return p


class Codebook:
    """midas_codebook handle over a (K, D) embedding matrix kept in HBM (float32 when lossless)."""

    def __init__(self, embeddings: torch.Tensor):
        ctx = _ctx(embeddings)
        emb = embeddings.contiguous()
        if emb.dtype == torch.float64:
            e32 = emb.float()
            # the TCN emits float32 codes cast to float64 (contrib/tcn_minkloc/tcn.py:148): store them
            # as float32 when that loses nothing, float64 otherwise
            if bool((e32.double() == emb).all()):
                emb = e32
        elif emb.dtype != torch.float32:
            emb = emb.float()
        self.emb = emb
        self.K, self.D = emb.shape
        self.ctx = ctx
        h = C.c_void_p()
        dtype = _lib.MIDAS_F32 if emb.dtype == torch.float32 else _lib.MIDAS_F64
        ctx.call("midas_codebook_create", self.K, self.D, _ptr(emb), dtype, C.byref(h))
        self.h = h

    def score(self, codes: torch.Tensor) -> torch.Tensor:
        """cos(code_b, C_k) for every row: (B, K) float64 (particle_filter.py:455-457, filter.py:213-215)."""
        codes = torch.atleast_2d(codes).to(self.emb.device, torch.float64).contiguous()
        if codes.shape[1] != self.D:
            raise MidasError(f"tactile code has {codes.shape[1]} dims, codebook has {self.D}")
        out = torch.empty((codes.shape[0], self.K), dtype=torch.float64, device=self.emb.device)
        self.ctx.call("midas_score", self.h, codes.shape[0], _ptr(codes), _ptr(out))
        return out

    def score_batch(self, codes: torch.Tensor) -> torch.Tensor:
        """(B, K) scores of B codes in one pass over the codebook on the matrix cores (float32 fma chains)."""
        codes = torch.atleast_2d(codes).to(self.emb.device, torch.float64).contiguous()
        if codes.shape[1] != self.D:
            raise MidasError(f"tactile code has {codes.shape[1]} dims, codebook has {self.D}")
        out = torch.empty((codes.shape[0], self.K), dtype=torch.float64, device=self.emb.device)
        self.ctx.call("midas_score_batch", self.h, codes.shape[0], _ptr(codes), _ptr(out))
        return out

    def __del__(self, _finalizing=sys.is_finalizing):  # pragma: no cover  # (bound at import: module globals are gone by then)
        if _finalizing():  # the process is going away: the HIP runtime may be gone already (its calls would abort, not raise)
            return
        try:
            if self.h:
                self.ctx.lib.midas_codebook_destroy(self.h)
                self.h = None
        except Exception:
            pass


class Tree:
    """Static KD-tree handle: dim 6 (float32 pose features) or dim 3 (float64 mesh vertices)."""

    def __init__(self, points: torch.Tensor):
        ctx = _ctx(points)
        dim = points.shape[1]
        if dim == 6:
            pts = points.float().contiguous()
        elif dim == 3:
            pts = points.double().contiguous()
        else:
            raise MidasError("Tree needs (K,6) float32 or (K,3) float64 points")
        self.dim, self.K, self.ctx = dim, pts.shape[0], ctx
        h = C.c_void_p()
        ctx.call("midas_tree_build", dim, self.K, _ptr(pts), C.byref(h))
        self.h = h

    def attach_mesh(self, mesh_tree: "Tree", cb_poses: torch.Tensor):
        """Per-codebook-entry mesh vertex lists: the prune of the fused step then rarely walks the mesh tree."""
        if self.dim != 6 or mesh_tree.dim != 3:
            raise MidasError("attach_mesh: self must be the 6-d codebook tree, mesh_tree the 3-d vertex tree")
        cb_poses = cb_poses.to(torch.float32).contiguous()
        self.ctx.call("midas_tree_attach_mesh", self.h, mesh_tree.h, _ptr(cb_poses))
        self._mesh = mesh_tree  # keep the mesh tree alive while the lists refer to it

    def export(self, what: str):
        """Host copy (numpy uint8 / float32 / int32) of one of the per-entry lists: "nbrs" (K, 513, 32) bytes, "rho_out" (K,),
        "twin" (K,), "vlist" (K, 257, 32) bytes.  For tests and inspection (midas_tree_export)."""
        import numpy as np
        code, shape, dt = {"nbrs": (0, (self.K, 513, 32), np.uint8), "rho_out": (1, (self.K,), np.float32),
                           "twin": (2, (self.K,), np.int32), "vlist": (3, (self.K, 257, 32), np.uint8)}[what]
        out = np.empty(shape, dtype=dt)
        self.ctx.call("midas_tree_export", self.h, code, C.c_void_p(out.ctypes.data), out.nbytes)
        return out

    def __del__(self, _finalizing=sys.is_finalizing):  # pragma: no cover  # (bound at import: module globals are gone by then)
        if _finalizing():  # the process is going away: the HIP runtime may be gone already (its calls would abort, not raise)
            return
        try:
            if self.h:
                self.ctx.lib.midas_tree_destroy(self.h)
                self.h = None
        except Exception:
            pass


def se3_feature(poses: torch.Tensor, w: float = 0.01) -> torch.Tensor:
    """R3_SE3 (tactile_tree/tactile_tree.py:73-77)."""
    poses = _poses(poses)
    out = torch.empty((poses.shape[0], 6), dtype=torch.float32, device=poses.device)
    _ctx(poses).call("midas_se3_feature", poses.shape[0], _ptr(poses), float(w), _ptr(out))
    return out


def nn6(tree: Tree, feat: torch.Tensor, hint: torch.Tensor | None = None, want_d2: bool = False):
    feat = feat.float().contiguous()
    n = feat.shape[0]
    idx = torch.empty(n, dtype=torch.int32, device=feat.device)
    d2 = torch.empty(n, dtype=torch.float32, device=feat.device) if want_d2 else None
    if hint is not None:
        hint = hint.to(torch.int32).contiguous()
    _ctx(feat).call("midas_nn6", tree.h, n, _ptr(feat), _ptr(hint), _ptr(idx), _ptr(d2))
    return (idx, d2) if want_d2 else idx


def knn6(tree: Tree, feat: torch.Tensor, k: int, want_d2: bool = False):
    """(N, k) int32 indices of the k nearest codebook entries per query, by (distance, index); exact (tactile_tree.py:50-52
    with n_neighbors = k)."""
    feat = feat.float().contiguous()
    n = feat.shape[0]
    idx = torch.empty((n, k), dtype=torch.int32, device=feat.device)
    d2 = torch.empty((n, k), dtype=torch.float32, device=feat.device) if want_d2 else None
    _ctx(feat).call("midas_knn6", tree.h, n, _ptr(feat), int(k), _ptr(idx), _ptr(d2))
    return (idx, d2) if want_d2 else idx


def nn6_stats(tree: Tree, feat: torch.Tensor, hint: torch.Tensor | None = None):
    """Diagnostic: (leaves visited, nodes visited) per query."""
    feat = feat.float().contiguous()
    n = feat.shape[0]
    leaves = torch.empty(n, dtype=torch.int32, device=feat.device)
    nodes = torch.empty(n, dtype=torch.int32, device=feat.device)
    if hint is not None:
        hint = hint.to(torch.int32).contiguous()
    _ctx(feat).call("midas_nn6_stats", tree.h, n, _ptr(feat), _ptr(hint), _ptr(leaves), _ptr(nodes))
    return leaves, nodes


def nn3_dist(tree: Tree, poses: torch.Tensor) -> torch.Tensor:
    poses = _poses(poses)
    dist = torch.empty(poses.shape[0], dtype=torch.float64, device=poses.device)
    _ctx(poses).call("midas_nn3", tree.h, poses.shape[0], _ptr(poses), _ptr(dist))
    return dist


def propagate(poses, odom, tn=None, rot=None, std_t=0.0, std_r=0.0, seed=0, step=0) -> torch.Tensor:
    """poses @ (odom @ Tn) (particle_filter.py:319-345,370-375); tn/rot None -> device Philox draws."""
    poses = _poses(poses)
    dev = poses.device
    odom = odom.to(dev, torch.float32).contiguous()
    out = torch.empty_like(poses)
    if tn is not None:
        tn = tn.to(dev, torch.float32).contiguous()
        rot = rot.to(dev, torch.float32).contiguous()
    _ctx(poses).call("midas_propagate", poses.shape[0], _ptr(poses), _ptr(out), _ptr(odom), _ptr(tn), _ptr(rot),
                     float(std_t), float(std_r), int(seed), int(step))
    return out


def check_poses(poses):
    poses = _poses(poses)
    flag = torch.empty(poses.shape[0], dtype=torch.uint8, device=poses.device)
    count = torch.zeros(1, dtype=torch.int32, device=poses.device)
    _ctx(poses).call("midas_check_poses", poses.shape[0], _ptr(poses), _ptr(flag), _ptr(count))
    return flag, count


def gather_f64(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    table = table.double().contiguous()
    idx = idx.to(torch.int32).contiguous()
    out = torch.empty(idx.shape[0], dtype=torch.float64, device=table.device)
    _ctx(table).call("midas_gather_f64", idx.shape[0], _ptr(table), _ptr(idx), _ptr(out))
    return out


def softmax_weights(x: torch.Tensor, softmax: bool = True) -> torch.Tensor:
    """get_similarity tail (particle_filter.py:459-468)."""
    x = x.double().contiguous()
    w = torch.empty_like(x)
    _ctx(x).call("midas_softmax", x.shape[0], _ptr(x), int(bool(softmax)), _ptr(w))
    return w


def prune_(w: torch.Tensor, dist: torch.Tensor, thr: float) -> torch.Tensor:
    """In place w *= !(dist > thr) (particle_filter.py:394-402); returns the kept count (device int32[1])."""
    if w.dtype != torch.float64 or not w.is_contiguous():
        raise MidasError("prune_ needs a contiguous float64 weight tensor (it is modified in place)")
    nvalid = torch.zeros(1, dtype=torch.int32, device=w.device)
    _ctx(w).call("midas_prune", w.shape[0], _ptr(w), _ptr(dist.double().contiguous()), float(thr), _ptr(nvalid))
    return nvalid


def cdf(w: torch.Tensor):
    w = w.double().contiguous()
    out = torch.empty_like(w)
    status = torch.zeros(1, dtype=torch.int32, device=w.device)
    _ctx(w).call("midas_cdf", w.shape[0], _ptr(w), _ptr(out), _ptr(status))
    return out, status


def resample_search(cdf_t: torch.Tensor, M: int, mode: int, u=None, u32: float = -1.0, seed=0, step=0) -> torch.Tensor:
    idx = torch.empty(M, dtype=torch.int32, device=cdf_t.device)
    if u is not None:
        u = u.to(cdf_t.device, torch.float64).contiguous()
    _ctx(cdf_t).call("midas_resample_search", cdf_t.shape[0], _ptr(cdf_t), M, mode, _ptr(u), float(u32), int(seed),
                     int(step), _ptr(idx))
    return idx


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """src[idx] along dim 0 (particle_filter.py:246-248; tactile_tree.py:54-58)."""
    src = src.contiguous()
    idx = idx.to(torch.int32).contiguous()
    out = torch.empty((idx.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    row_bytes = src.element_size()
    for s in src.shape[1:]:
        row_bytes *= s
    if idx.shape[0] and row_bytes:
        _ctx(src).call("midas_gather_rows", idx.shape[0], _ptr(idx), _ptr(src), _ptr(out), row_bytes)
    return out


def rmse(poses: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    poses = _poses(poses)
    gt = gt.to(poses.device, torch.float32).contiguous()
    out = torch.empty(2, dtype=torch.float64, device=poses.device)
    _ctx(poses).call("midas_rmse", poses.shape[0], _ptr(poses), _ptr(gt), _ptr(out))
    return out


def cluster_centers(poses: torch.Tensor, weights: torch.Tensor, labels: torch.Tensor, label_values: torch.Tensor):
    """K9: (centres (C,4,4) f32, stds (C,3) f32, members (C,) i64) of the clusters `label_values` (midas_cluster_centers)."""
    poses = _poses(poses)
    dev = poses.device
    labels = labels.to(dev, torch.int64).contiguous()
    label_values = label_values.to(dev, torch.int64).contiguous()
    weights = weights.to(dev).contiguous()
    if weights.dtype not in (torch.float32, torch.float64):
        weights = weights.double()
    C = int(label_values.shape[0])
    centers = torch.empty((C, 4, 4), dtype=torch.float32, device=dev)
    stds = torch.empty((C, 3), dtype=torch.float32, device=dev)
    counts = torch.empty((C,), dtype=torch.int64, device=dev)
    w64 = _ptr(weights) if weights.dtype == torch.float64 else None
    w32 = _ptr(weights) if weights.dtype == torch.float32 else None
    _ctx(poses).call("midas_cluster_centers", poses.shape[0], _ptr(poses), w64, w32, _ptr(labels), C, _ptr(label_values),
                     _ptr(centers), _ptr(stds), _ptr(counts))
    return centers, stds, counts


def dbscan(poses: torch.Tensor, eps: float = 1e-2, min_samples: int = -1):
    """cluster_particles(method="euclidean") labels (particle_filter.py:208-217): DBSCAN of the translations on the device.
    min_samples < 0 -> N // 5.  Returns (labels int32 (N,), info int32 (2,) = [clusters, limit flag])."""
    poses = _poses(poses)
    labels = torch.empty(poses.shape[0], dtype=torch.int32, device=poses.device)
    info = torch.zeros(2, dtype=torch.int32, device=poses.device)
    _ctx(poses).call("midas_dbscan", poses.shape[0], _ptr(poses), float(eps), int(min_samples), _ptr(labels), _ptr(info))
    return labels, info


def dbscan_points(points: torch.Tensor, eps: float = 1e-2, min_samples: int = -1):
    """DBSCAN labels of N points in 2 .. 6 dimensions (cluster_particles(method="logmap"), particle_filter.py:218-223: the
    6-d SE(3) logarithms), all pairs on the device in float64 - sklearn's predicate and numbering.
    Returns (labels int32 (N,), info int32 (2,) = [clusters, spread steps or -1])."""
    pts = points.to(torch.float64).contiguous()
    if not pts.is_cuda:
        raise MidasError("midastouch_amd kernels need tensors on a HIP device; there is no CPU fallback")
    labels = torch.empty(pts.shape[0], dtype=torch.int32, device=pts.device)
    info = torch.zeros(2, dtype=torch.int32, device=pts.device)
    _ctx(pts).call("midas_dbscan_points", pts.shape[0], int(pts.shape[1]), _ptr(pts), float(eps), int(min_samples), _ptr(labels), _ptr(info))
    return labels, info


def anneal_select(weights: torch.Tensor, mode: int, k: int, ties: str = "index", info: torch.Tensor = None) -> torch.Tensor:
    """Index list of the annealed particle set (particle_filter.py:421-446): mode 1 = without the k smallest weights (order
    kept), mode 2 = everybody followed by the k largest (largest first).  ties: "index" = to the smaller index (torch's CUDA
    kernel), "aten_cpu" = the members / the order `torch.topk` returns on the CPU (topk_aten.hip).  info: optional int32[1]
    the call adds its depth-limit fallbacks to."""
    w = weights.double().contiguous()
    n = w.shape[0]
    src = torch.empty(n + (k if mode == 2 else 0), dtype=torch.int32, device=w.device)
    rule = {"index": _lib.TOPK_TIES_INDEX, "aten_cpu": _lib.TOPK_TIES_ATEN_CPU}[ties]
    _ctx(w).call("midas_anneal_select_ties", n, _ptr(w), int(mode), int(k), rule, _ptr(src), _ptr(info))
    return src[:n - k] if mode == 1 else src


def topn_pose_error(scores: torch.Tensor, row0: int, n: int, feat: torch.Tensor, want_idx: bool = False):
    """Per row of `scores` (B, K) f64 - row b = similarities of entry row0 + b - the best pose error among its n
    best-scoring entries, diagonal zeroed (midas_topn_pose_error; eval/single_touch_test.py:35-73)."""
    scores = scores.to(torch.float64).contiguous()
    feat = feat.to(scores.device, torch.float64).contiguous()
    B, K = scores.shape
    err = torch.empty((B,), dtype=torch.float64, device=scores.device)
    idx = torch.empty((B, n), dtype=torch.int32, device=scores.device) if want_idx else None
    _ctx(scores).call("midas_topn_pose_error", B, K, _ptr(scores), int(row0), int(n), _ptr(feat), int(feat.shape[1]), _ptr(err),
                      _ptr(idx))
    return (err, idx) if want_idx else err
