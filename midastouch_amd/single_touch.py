"""Single-touch evaluation of a codebook: `eval/single_touch_test.py:35-91` on the device.

`top_n_error` is the reference's K x K self-similarity of the embeddings followed by a per-row top-25 and the best pose
error among them.  Here the similarity rows come from the scoring kernels tile by tile (exact float64 GEMV rows by default;
`fast=True`: the self-similarity as a float32 GEMM on the matrix cores, `midas_selfsim_topn` - panels of `panel_rows`
queries against all K entries) and the selection kernel (`midas_topn_pose_error`) consumes each tile in one pass - the
K x K matrix (20 GB at K = 50 k) never exists.
The heat-map of `filter/filter.py:213-215` is `particle_filter.get_similarity(code, codebook.get_embeddings(), softmax=False)`.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops

NUM_NEIGHBORS = 25  # single_touch_test.py:32


def top_n_error(embeddings: torch.Tensor, poses: torch.Tensor, n: int = NUM_NEIGHBORS, fast: bool = False,
                tile: int = 256, want_idx: bool = False, panel_rows: int = 4096):
    """(K,) float64: for every codebook entry the smallest |pose_j - pose_i| among its n most similar entries
    (diagonal similarity set to 0 like `np.fill_diagonal(C, 0)`).  embeddings (K, D) and poses (K, d) on a HIP device."""
    emb = embeddings if isinstance(embeddings, torch.Tensor) else torch.as_tensor(embeddings)
    if not emb.is_cuda:
        raise ops.MidasError("top_n_error needs the embeddings on a HIP device; there is no CPU fallback")
    cb = ops.Codebook(emb)
    feat = torch.as_tensor(poses).to(emb.device, torch.float64).reshape(emb.shape[0], -1)
    K = emb.shape[0]
    out = torch.empty((K,), dtype=torch.float64, device=emb.device)
    idx_all = torch.empty((K, n), dtype=torch.int32, device=emb.device) if want_idx else None
    if fast and cb.emb.dtype == torch.float32 and cb.D % 32 == 0:
        # the whole K x K x D self-similarity as a float32 GEMM on the matrix cores (midas_selfsim_topn), panel by panel
        cb.ctx.call("midas_selfsim_topn", cb.h, int(n), ops._ptr(feat), int(feat.shape[1]), int(panel_rows), ops._ptr(out),
                    ops._ptr(idx_all))
        return (out, idx_all) if want_idx else out
    tile = min(int(tile), 64) if fast else int(tile)
    for i0 in range(0, K, tile):
        q = cb.emb[i0:i0 + tile].to(torch.float64)
        scores = cb.score_batch(q) if fast else cb.score(q)
        r = ops.topn_pose_error(scores, i0, n, feat, want_idx=want_idx)
        if want_idx:
            out[i0:i0 + tile], idx_all[i0:i0 + tile] = r
        else:
            out[i0:i0 + tile] = r
    return (out, idx_all) if want_idx else out


def get_random_error(poses, n: int = NUM_NEIGHBORS, rng=None) -> float:
    """Mean best error of n random picks per entry (single_touch_test.py:76-91); host arithmetic, numpy draws."""
    poses = np.asarray(torch.as_tensor(poses).cpu(), dtype=np.float64).reshape(len(poses), -1)
    rng = np.random if rng is None else rng
    N = poses.shape[0]
    err = np.zeros(N)
    for i in range(N):
        pred = rng.choice(N, size=n)
        err[i] = np.min(np.linalg.norm(poses[pred] - poses[i], axis=1))
    return float(np.mean(err))
