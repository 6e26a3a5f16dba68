"""top_n_error against a numpy restatement of eval/single_touch_test.py:35-73."""
import numpy as np
import pytest
import torch

from midastouch_amd.synthetic import make_codebook


def ref_top_n_error(emb, poses, n):
    """The reference's arithmetic (sklearn cosine_similarity = normalised rows, float64), stable selection."""
    X = emb.astype(np.float64)
    X = X / np.linalg.norm(X, axis=1, keepdims=True)
    C = X @ X.T
    np.fill_diagonal(C, 0)
    out = np.zeros(len(X))
    best = np.zeros((len(X), n), dtype=np.int64)
    for i in range(len(X)):
        order = np.lexsort((np.arange(len(X)), -C[i]))[:n]  # value descending, index ascending
        best[i] = order
        out[i] = np.min(np.linalg.norm(poses[order] - poses[i], axis=1))
    return out, best, C


@pytest.mark.gpu
@pytest.mark.parametrize("K,n,tile", [(700, 25, 256), (2500, 25, 64), (3000, 7, 1000), (20, 25, 256)])
def test_top_n_error_matches_reference_arithmetic(K, n, tile):
    from midastouch_amd.single_touch import top_n_error
    dev = torch.device("cuda", 0)
    cb = make_codebook(K=K, D=128, seed=1200, mesh_points=2000)
    poses = cb.poses[:, :3, 3].astype(np.float64)
    ref, best, C = ref_top_n_error(cb.embeddings, poses, min(n, K))
    err, idx = top_n_error(torch.as_tensor(cb.embeddings).to(dev), torch.as_tensor(poses).to(dev), n=n, tile=tile, want_idx=True)
    idx = idx.cpu().numpy()
    if K >= n:
        # the selected sets agree wherever the n-th and (n+1)-th similarities differ by more than rounding
        srt = -np.sort(-C, axis=1)
        clear = (srt[:, n - 1] - srt[:, n]) > 1e-12 if K > n else np.ones(K, bool)
        assert clear.mean() > 0.99
        same = np.array([set(idx[i]) == set(best[i]) for i in range(K)])
        assert same[clear].all()
        np.testing.assert_allclose(err.cpu().numpy()[clear], ref[clear], rtol=1e-12, atol=1e-15)
    else:  # fewer entries than n: everything is selected, the diagonal (similarity 0, distance 0) included
        assert (idx[:, K:] == -1).all()
        assert np.allclose(err.cpu().numpy(), 0.0)


@pytest.mark.gpu
def test_top_n_error_fast_path_close_to_exact():
    from midastouch_amd.single_touch import top_n_error
    dev = torch.device("cuda", 0)
    cb = make_codebook(K=3000, D=256, seed=1201, mesh_points=2000)
    poses = torch.as_tensor(cb.poses[:, :3, 3]).to(dev)
    emb = torch.as_tensor(cb.embeddings).to(dev)
    exact = top_n_error(emb, poses)
    fast = top_n_error(emb, poses, fast=True)
    # float32 accumulation can swap entries at rank 25/26; the metric moves on a handful of rows at most
    assert (exact != fast).float().mean() < 0.01
    assert abs(float(exact.mean()) - float(fast.mean())) < 1e-3 * float(exact.mean())


def test_random_error_host():
    from midastouch_amd.single_touch import get_random_error
    poses = np.random.default_rng(0).uniform(-0.1, 0.1, (200, 3))
    e = get_random_error(poses, n=25, rng=np.random.default_rng(1))
    assert 0.0 < e < 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("K,D", [(3000, 256), (1000, 128), (700, 512), (130, 64)])
def test_selfsim_panel_equals_batched_scorer_bit_for_bit(K, D, oracle):
    """k_selfsim_mfma (the K x K x D float32 GEMM of top_n_error on the matrix cores): every dot product equals the oracle's
    float32 fma chain in the matrix-core order (mo_score_batch_f32's numerator) bit for bit - whole tiles, ragged edges,
    panels that start in the middle of the codebook."""
    from midastouch_amd import ops
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(K + D)
    E = rng.standard_normal((K, D)).astype(np.float32)
    cbk = ops.Codebook(torch.as_tensor(E).to(dev))
    ldo = -(-K // 128) * 128
    for i0, R in ((0, min(K, 300)), (K // 3, min(K - K // 3, 129))):
        panel = torch.zeros((-(-R // 128) * 128, ldo), dtype=torch.float32, device=dev)
        cbk.ctx.call("midas_selfsim_panel", cbk.h, i0, R, ops._ptr(panel), ldo)
        got = panel[:R, :K].cpu().numpy()
        # the oracle's chain: for c (16 d-values), s, g: acc = fmaf(E_j[d], E_i[d], acc), d = 16 c + 4 g + s
        order = np.array([16 * c + 4 * g + s for c in range(D // 16) for s in range(4) for g in range(4)])
        rows = rng.choice(R, size=min(R, 12), replace=False)
        for r in rows:
            acc = np.zeros(K, dtype=np.float32)
            q = E[i0 + r]
            for dd in order:  # float32 fma emulated in float64 (the product of two float32 is exact there, one rounding)
                acc = (E[:, dd].astype(np.float64) * np.float64(q[dd]) + acc.astype(np.float64)).astype(np.float32)
            assert np.array_equal(got[r], acc), (K, D, i0, r)


@pytest.mark.gpu
def test_top_n_error_gemm_path_matches_reference_arithmetic():
    """fast=True (midas_selfsim_topn) against the numpy restatement of the reference's arithmetic: the selected sets agree
    wherever the n-th and (n+1)-th similarities are apart by more than float32 accumulation error."""
    from midastouch_amd.single_touch import top_n_error
    dev = torch.device("cuda", 0)
    for K, D, n, rows in ((2500, 256, 25, 512), (900, 128, 7, 128)):
        cb = make_codebook(K=K, D=D, seed=1300 + K, mesh_points=2000)
        poses = cb.poses[:, :3, 3].astype(np.float64)
        ref, best, C = ref_top_n_error(cb.embeddings, poses, n)
        err, idx = top_n_error(torch.as_tensor(cb.embeddings).to(dev), torch.as_tensor(poses).to(dev), n=n, fast=True, want_idx=True, panel_rows=rows)
        idx = idx.cpu().numpy()
        srt = -np.sort(-C, axis=1)
        clear = (srt[:, n - 1] - srt[:, n]) > 1e-5
        assert clear.mean() > 0.9
        same = np.array([set(idx[i]) == set(best[i]) for i in range(K)])
        assert same[clear].all()
        np.testing.assert_allclose(err.cpu().numpy()[clear], ref[clear], rtol=1e-12, atol=1e-15)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["regular", "ties", "small_n", "zero_row", "wide"])
def test_selection_forms_of_the_gemm_path_agree(case, monkeypatch):
    """The register-resident selection of a panel row (k_topn_dots_rows: float32 screen, float64 decision among the
    candidates) against the streaming kernel on the same panels (MIDAS_TOPN_STREAM=1): the same indices in the same order and
    the same errors - also where more rows tie than the candidate buffer holds (the row then goes through the streaming
    form inside the kernel), where a row's own norm is zero, and on rows wider than one load round."""
    from midastouch_amd.single_touch import top_n_error
    dev = torch.device("cuda", 0)
    K, D, n, rows = {"regular": (2500, 256, 25, 512), "ties": (3000, 128, 25, 1024), "small_n": (900, 128, 7, 128),
                     "zero_row": (1500, 128, 25, 256), "wide": (20000, 64, 25, 2048)}[case]
    rng = np.random.default_rng(77)
    if case == "wide":
        E = rng.standard_normal((K, D)).astype(np.float32)
        poses = rng.uniform(-0.1, 0.1, (K, 3))
    else:
        cb = make_codebook(K=K, D=D, seed=1400 + K, mesh_points=2000)
        E, poses = cb.embeddings.astype(np.float32).copy(), cb.poses[:, :3, 3].astype(np.float64)
    if case == "ties":
        E[500:2300] = E[499]       # 1801 identical rows: every one of them ties at cosine 1 with the others
    if case == "zero_row":
        E[7] = 0.0                 # every dot of this row is 0 (the norms are clamped away from 0): K scores tie at 0
        E[100:140] = E[99]
    emb, ps = torch.as_tensor(E).to(dev), torch.as_tensor(poses).to(dev)
    monkeypatch.setenv("MIDAS_TOPN_STREAM", "1")
    e0, i0 = top_n_error(emb, ps, n=n, fast=True, want_idx=True, panel_rows=rows)
    monkeypatch.setenv("MIDAS_TOPN_STREAM", "0")
    e1, i1 = top_n_error(emb, ps, n=n, fast=True, want_idx=True, panel_rows=rows)
    i0, i1 = i0.cpu().numpy(), i1.cpu().numpy()
    assert np.array_equal(i0, i1)
    assert np.array_equal(e0.cpu().numpy(), e1.cpu().numpy(), equal_nan=True)
    if case == "ties":
        assert (i1[600] == np.r_[499:600, 601:2300][:n]).all()   # value descending, index ascending
    if case == "zero_row":
        assert (i1[7] == np.arange(n)).all()
