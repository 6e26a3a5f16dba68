#!/usr/bin/env python3
"""Round-2 golden fixtures, produced by running the REAL reference functions (build container only; imports
/root/reference with the same empty import shims as tools/gen_goldens.py - no shimmed function is called).

  G2b  particle_filter.resampler at N = 100 000   modules/particle_filter.py:230-307
       12 weight sets x {weighted_random, low_var}.  The weights are rebuilt by the test from integer recipes (exact
       float64 arithmetic, no libm), the draws from the stored torch seed; the fixture holds the SHA-256 of the
       reference's index array and its first / last 64 entries - the regime where summation-order effects on the CDF
       would first show (SURVEY.md section 7 hard part 2).
  G9   particle_filter.cluster_particles           modules/particle_filter.py:208-228 (sklearn DBSCAN, eps 1e-2,
       min_samples N/5) on synthetic particle sets: labels.
  G12  eval/single_touch_test.top_n_error          eval/single_touch_test.py:35-73 on a small synthetic codebook.

A fixture is data only: inputs (or their recipe) + the reference's outputs.
Usage:  python tools/gen_goldens_r2.py
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
from gen_goldens import import_reference, marker_poses, new_pf, save  # noqa: E402


def recipe_weights(n: int, kind: str, seed: int) -> np.ndarray:
    """Weights from integer hashes only - bit-identical on every platform.  (The same function lives in
    tests/test_oracle_golden.py; the fixture's `w_sha` guards against the two drifting apart.)"""
    i = np.arange(n, dtype=np.uint64)
    h = (i + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
    h ^= h >> np.uint64(29)
    h *= np.uint64(0xBF58476D1CE4E5B9)
    h ^= h >> np.uint64(32)
    frac = ((h >> np.uint64(11)) & np.uint64((1 << 20) - 1)).astype(np.float64) / float(1 << 20)  # exact
    expo = (h & np.uint64(63)).astype(np.int64)
    if kind == "flat":        # one binade: a converged softmax
        w = 1.0 + frac
    elif kind == "peaky":     # 40 binades: a sharp softmax
        w = np.ldexp(1.0 + frac, -(expo % 40))
    elif kind == "masked":    # 60 % pruned to exactly zero
        w = np.ldexp(1.0 + frac, -(expo % 8)) * ((h >> np.uint64(40)) % np.uint64(10) < np.uint64(4))
    elif kind == "dupes":     # few distinct values: particles sharing a codebook entry share a weight
        w = np.ldexp(1.0, -((expo % 12).astype(np.int64))) * (1.0 + (expo % 3) / 4.0)
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(w, dtype=np.float64)


def g2b_resampler_100k(pfm):
    pf = new_pf(pfm)
    n = 100000
    out = {"N": np.int64(n)}
    cases = [(k, s) for k in ("flat", "peaky", "masked", "dupes") for s in (1, 2, 3)]
    for ci, (kind, s) in enumerate(cases):
        w = recipe_weights(n, kind, 1000 * s + ci)
        out[f"c{ci}_kind"], out[f"c{ci}_wseed"] = np.str_(kind), np.int64(1000 * s + ci)
        out[f"c{ci}_w_sha"] = np.str_(hashlib.sha256(w.tobytes()).hexdigest())
        for mode in ("weighted_random", "low_var"):
            seed = 7000 + 10 * ci + (mode == "low_var")
            P = pfm.Particles(marker_poses(n), torch.tensor(w), torch.arange(n, dtype=torch.float32))
            torch.manual_seed(seed)
            r = pf.resampler(P, resample=mode)
            idx = r.poses[:, 0, 3].numpy().astype(np.int32)
            assert np.array_equal(r.weights.numpy(), w[idx])
            out[f"c{ci}_{mode}_seed"] = np.int64(seed)
            out[f"c{ci}_{mode}_sha"] = np.str_(hashlib.sha256(np.ascontiguousarray(idx).tobytes()).hexdigest())
            out[f"c{ci}_{mode}_head"], out[f"c{ci}_{mode}_tail"] = idx[:64].copy(), idx[-64:].copy()
    out["ncases"] = np.int64(len(cases))
    save("g2b_resampler_100k", **out)


def g9_dbscan(pfm):
    pf = new_pf(pfm)
    rng = np.random.default_rng(909)
    out = {}

    def run(tag, X, eps=None):
        X = np.ascontiguousarray(X, dtype=np.float32)
        P = torch.eye(4)[None].repeat(len(X), 1, 1).clone()
        P[:, :3, 3] = torch.tensor(X)
        parts = pfm.Particles(P, torch.ones(len(X), dtype=torch.float64), torch.zeros(len(X)))
        res = pf.cluster_particles(parts) if eps is None else pf.cluster_particles(parts, eps=eps)
        lab = res.labels.numpy()
        out[f"{tag}_X"], out[f"{tag}_labels"] = X, lab.astype(np.int32)
        out[f"{tag}_eps"] = np.float64(1e-2 if eps is None else eps)
        print(f"  g9 {tag}: N={len(X)} labels {dict(zip(*np.unique(lab, return_counts=True)))}")

    # two blobs + background: two clusters, border points, noise
    run("two", np.concatenate([rng.normal(0.0, 0.004, (1200, 3)), rng.normal(0.03, 0.004, (1100, 3)),
                               rng.uniform(-0.05, 0.08, (200, 3))]))
    # one converged cluster
    run("one", rng.normal(0.01, 0.002, (1500, 3)))
    # nothing dense enough: all noise
    run("noise", rng.uniform(-0.1, 0.1, (2000, 3)))
    # three blobs whose halos overlap (border points within eps of two clusters take the smaller label), shuffled so
    # that cluster numbers follow the first core point in index order
    bridge = np.stack([rng.uniform(0.008, 0.044, 120), rng.normal(0, 0.002, 120), rng.normal(0, 0.002, 120)], axis=1)
    X = np.concatenate([rng.normal([0, 0, 0], 0.003, (700, 3)), rng.normal([0.026, 0, 0], 0.003, (700, 3)),
                        rng.normal([0.052, 0.002, 0], 0.003, (700, 3)), bridge, rng.uniform(-0.02, 0.07, (60, 3))])
    run("three", X[rng.permutation(len(X))])
    # points on a lattice with spacing exactly eps / 2 in float32: many distances at the threshold
    g = (np.arange(5, dtype=np.float32) * np.float32(0.005))
    L = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    run("lattice", L, eps=0.01)
    # surface-walk shaped set at a larger eps (the eps argument travels)
    run("eps2", np.concatenate([rng.normal(0, 0.008, (900, 3)), rng.normal(0.06, 0.008, (900, 3))]), eps=0.02)
    save("g9_dbscan", **out)


def g12_topn():
    for name in ("seaborn", "midastouch.viz", "midastouch.viz.helpers", "GPUtil", "git", "cv2", "ffmpeg", "pyvista", "open3d"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["midastouch.viz.helpers"].viz_embedding_TSNE = None
    # modules/misc.py:37 locates the checkout with gitpython at import time: the shim answers with the reference path
    sys.modules["git"].Repo = lambda *a, **k: types.SimpleNamespace(working_tree_dir="/root/reference")
    from midastouch.eval import single_touch_test as stt
    from midastouch_amd.synthetic import make_codebook

    out = {}
    for tag, (K, D, n) in {"a": (600, 128, 25), "b": (900, 256, 25), "c": (300, 64, 7)}.items():
        cb = make_codebook(K=K, D=D, seed=1200 + K, mesh_points=2000)
        poses = cb.poses[:, :3, 3].astype(np.float64)
        err = stt.top_n_error(cb.embeddings.astype(np.float64), poses, n=n)
        # inputs by recipe (midastouch_amd.synthetic.make_codebook is deterministic), guarded by a digest
        out[f"{tag}_K"], out[f"{tag}_D"], out[f"{tag}_n"], out[f"{tag}_seed"] = np.int64(K), np.int64(D), np.int64(n), np.int64(1200 + K)
        out[f"{tag}_emb_sha"] = np.str_(hashlib.sha256(np.ascontiguousarray(cb.embeddings, dtype=np.float32).tobytes()).hexdigest())
        out[f"{tag}_err"] = err
    save("g12_topn", **out)


def main():
    torch.set_num_threads(1)
    pfm, _ = import_reference()
    g2b_resampler_100k(pfm)
    g9_dbscan(pfm)
    g12_topn()


if __name__ == "__main__":
    main()
