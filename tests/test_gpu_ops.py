"""Parity of every HIP kernel group against the CPU oracle, through the C ABI.  Needs an MI355X.

Bars: float32 pose/feature arithmetic and every index bit-exact (the kernels restate the oracle's
arithmetic spec); since round 3 the float64 scores (fixed summation order of the dot products) and the softmax
numerators (spec exponential) are bit-exact too; weights against the reference's goldens 1e-12; resample indices bit-exact.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def ops():
    from midastouch_amd import ops as o
    return o


@pytest.fixture(scope="module")
def cb():
    from midastouch_amd.synthetic import make_codebook
    return make_codebook(K=5000, D=256, seed=1000)


def T(a, dev, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a)).to(dev)
    return t.to(dtype) if dtype is not None else t


def _rand_poses(n, seed, scale=0.1):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    P = np.zeros((n, 4, 4), dtype=np.float32)
    P[:, :3, :3] = Rotation.random(n, random_state=seed).as_matrix()
    P[:, :3, 3] = rng.uniform(-scale, scale, size=(n, 3))
    P[:, 3, 3] = 1
    return P


def test_propagate_host_noise_bit_exact(dev, ops, oracle, golden):
    g = golden("g3_motion")
    for tag in ("sim", "mc", "mul3", "big"):
        P, odom, tn, rot = g[f"{tag}_poses"], g[f"{tag}_odom"], g[f"{tag}_tn"], g[f"{tag}_rot"]
        out = ops.propagate(T(P, dev), T(odom, dev), T(tn, dev), T(rot, dev)).cpu().numpy()
        ref = oracle.propagate(P, odom, tn, rot)
        assert np.array_equal(out, ref), tag
        # and against the reference's own result (torch sin/cos + BLAS order): float tolerance
        np.testing.assert_allclose(out, g[f"{tag}_new_poses"], rtol=0, atol=2e-6)


def test_propagate_philox_bit_exact(dev, ops, oracle):
    P = _rand_poses(5000, 3)
    odom = _rand_poses(1, 4, 1e-3)[0]
    out = ops.propagate(T(P, dev), T(odom, dev), None, None, std_t=2e-4, std_r=0.5, seed=4000, step=17).cpu().numpy()
    tn, rot = oracle.philox_noise(5000, 4000, 17, 2e-4, 0.5)
    ref = oracle.propagate(P, odom, tn, rot)
    assert np.array_equal(out, ref)


def test_se3_feature_bit_exact(dev, ops, oracle, cb):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(8)
    P = np.concatenate([cb.poses, _rand_poses(3000, 5)])
    # add near-identity and near-pi rotations
    near0 = _rand_poses(500, 6)
    near0[:, :3, :3] = Rotation.from_rotvec(rng.standard_normal((500, 3)) * 1e-3).as_matrix()
    ax = rng.standard_normal((500, 3))
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    nearpi = _rand_poses(500, 7)
    nearpi[:, :3, :3] = Rotation.from_rotvec(ax * (np.pi - rng.uniform(0, 0.15, (500, 1)))).as_matrix()
    P = np.concatenate([P, near0, nearpi]).astype(np.float32)
    out = ops.se3_feature(T(P, dev)).cpu().numpy()
    assert np.array_equal(out, oracle.R3_SE3(P))


def test_nn6_exact(dev, ops, oracle, cb):
    feat = oracle.R3_SE3(cb.poses)
    tree = ops.Tree(T(feat, dev))
    rng = np.random.default_rng(9)
    q_near = feat[rng.integers(0, cb.K, 4000)] + (rng.standard_normal((4000, 6)) * 3e-4).astype(np.float32)
    q_far = oracle.R3_SE3(_rand_poses(2000, 10, 0.15))
    q = np.concatenate([q_near, q_far, feat[:100]]).astype(np.float32)
    ref_idx, ref_d2 = oracle.nn6(q, feat)
    idx, d2 = ops.nn6(tree, T(q, dev), want_d2=True)
    assert np.array_equal(idx.cpu().numpy(), ref_idx)
    assert np.array_equal(d2.cpu().numpy(), ref_d2)
    # a hint (right, wrong or out of range) never changes the answer
    hint = ref_idx.copy()
    hint[::3] = rng.integers(0, cb.K, len(hint[::3]))
    hint[::7] = -1
    idx2 = ops.nn6(tree, T(q, dev), hint=T(hint, dev))
    assert np.array_equal(idx2.cpu().numpy(), ref_idx)


def test_nn6_pivot_switch_across_pi_cut(dev, ops, oracle):
    """Particles whose rotation angle passes pi reappear 63 mm-equivalents away from their ancestor's nearest entry in the
    feature space of R3_SE3 (tactile_tree/tactile_tree.py:73-77: 0.01 log R flips sign at the cut).  The hinted scan
    continues from the hinted entry's twin: same exact answer as the brute force, and neither the whole 512-record list
    nor the tree search is needed for them (what made the frames after a wide start slow)."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(31)
    K = 6000
    ax = rng.standard_normal((K, 3))
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    ang = np.where(rng.random(K) < 0.5, np.pi - rng.uniform(0, 0.3, K), rng.uniform(0, np.pi, K))
    P = np.zeros((K, 4, 4), dtype=np.float32)
    P[:, :3, :3] = Rotation.from_rotvec(ax * ang[:, None]).as_matrix()
    P[:, :3, 3] = rng.uniform(-0.02, 0.02, (K, 3))
    P[:, 3, 3] = 1
    feat = oracle.R3_SE3(P)
    tree = ops.Tree(T(feat, dev))
    src = np.flatnonzero(ang > np.pi - 0.03)
    src = np.concatenate([src, rng.integers(0, K, 2000)])
    Q = P[src].copy()
    dR = Rotation.from_rotvec(rng.standard_normal((len(src), 3)) * 0.02).as_matrix().astype(np.float32)
    Q[:, :3, :3] = Q[:, :3, :3] @ dR
    Q[:, :3, 3] += (rng.standard_normal((len(src), 3)) * 2e-4).astype(np.float32)
    q = oracle.R3_SE3(Q)
    ref, _ = oracle.nn6(q, feat)
    far = np.linalg.norm(q - feat[src], axis=1) > 0.02  # crossed the cut: far from the hinted entry
    assert far.sum() > 20
    idx = ops.nn6(tree, T(q, dev), hint=T(src.astype(np.int32), dev))
    assert np.array_equal(idx.cpu().numpy(), ref)
    leaves, nodes = ops.nn6_stats(tree, T(q, dev), T(src.astype(np.int32), dev))
    leaves, nodes = leaves.cpu().numpy(), nodes.cpu().numpy()
    assert np.all(leaves[far] == 0), "cut-crossing particles fell back to the tree search"
    # nodes = -(1 + records scanned by the lane itself) for lanes the solo scan certified (at most NN_SOLO = 32 records)
    solo = nodes[far] < 0
    assert solo.mean() > 0.9, "cut-crossing particles should certify from the twin's first records"


def test_nn6_ties_and_small_trees(dev, ops, oracle):
    rng = np.random.default_rng(11)
    for K in (1, 2, 7, 8, 9, 17, 100):
        pts = rng.standard_normal((K, 6)).astype(np.float32)
        pts = np.concatenate([pts, pts[: max(1, K // 2)]])  # duplicated points: ties -> smallest index
        q = np.concatenate([pts, rng.standard_normal((50, 6)).astype(np.float32)])
        tree = ops.Tree(T(pts, dev))
        ref, _ = oracle.nn6(q, pts)
        assert np.array_equal(ops.nn6(tree, T(q, dev)).cpu().numpy(), ref), K


def test_knn6_exact_order_and_ties(dev, ops, oracle):
    """k nearest by (distance, index): equal to the oracle's brute force, duplicates included; column 0 == midas_nn6."""
    rng = np.random.default_rng(21)
    for K, k in ((5, 5), (40, 3), (700, 8), (5000, 64), (3000, 17)):
        pts = rng.standard_normal((K, 6)).astype(np.float32)
        pts[K // 2:] = pts[: K - K // 2] if K >= 40 else pts[K // 2:]  # duplicated points: ties -> smaller index first
        q = np.concatenate([pts[:30], rng.standard_normal((70, 6)).astype(np.float32)])
        tree = ops.Tree(T(pts, dev))
        idx, d2 = ops.knn6(tree, T(q, dev), k, want_d2=True)
        ref_i, ref_d = oracle.knn6(q, pts, k)
        assert np.array_equal(idx.cpu().numpy(), ref_i), (K, k)
        assert np.array_equal(d2.cpu().numpy(), ref_d), (K, k)
        assert np.array_equal(idx[:, 0].cpu().numpy(), ops.nn6(tree, T(q, dev)).cpu().numpy())
    with pytest.raises(Exception):
        ops.knn6(tree, T(q, dev), 65)


def test_nn3_dist_bit_exact(dev, ops, oracle, cb, golden):
    verts = cb.mesh_vertices
    tree = ops.Tree(T(verts, dev))
    rng = np.random.default_rng(12)
    P = _rand_poses(4000, 13)
    P[:, :3, 3] = (verts[rng.integers(0, len(verts), 4000)] + rng.standard_normal((4000, 3)) * 2e-3).astype(np.float32)
    dist = ops.nn3_dist(tree, T(P, dev)).cpu().numpy()
    assert np.array_equal(dist, oracle.nn3_dist(P, verts))
    # reference golden (sklearn KDTree distances)
    g = golden("g4_prune")
    tree2 = ops.Tree(T(g["verts"], dev))
    for tag in ("near", "far", "thr"):
        pos = g[f"{tag}_pos"]
        Q = np.repeat(np.eye(4, dtype=np.float32)[None], len(pos), 0)
        Q[:, :3, 3] = pos
        d = ops.nn3_dist(tree2, T(Q, dev))
        np.testing.assert_allclose(d.cpu().numpy(), g[f"{tag}_dist"], rtol=1e-13)
        w = T(g[f"{tag}_w_in"], dev).clone()
        kept = ops.prune_(w, d, float(g[f"{tag}_thr"]))
        assert np.array_equal(w.cpu().numpy(), g[f"{tag}_w_out"])
        assert (int(kept.item()) == 0) == bool(g[f"{tag}_drifted"])


@pytest.mark.parametrize("D", [256, 512, 128, 100, 1024])
def test_score_codebook(dev, ops, oracle, D):
    rng = np.random.default_rng(D)
    K = 3001
    E = rng.standard_normal((K, D)).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    E[5] = 0.0  # eps clamp
    code = rng.standard_normal(D)
    code /= np.linalg.norm(code)
    ref = oracle.score_codebook(E, code)
    E64 = E.astype(np.float64) * (1.0 + 1e-9)  # not float32-representable -> stays float64 in HBM
    for emb, want in ((T(E, dev), ref), (T(E, dev).double(), ref), (T(E64, dev), oracle.score_codebook(E64, code))):
        cbk = ops.Codebook(emb)
        assert cbk.emb.dtype == (torch.float64 if emb is not None and want is not ref else torch.float32)
        s = cbk.score(T(code, dev)).cpu().numpy()[0]
        assert np.array_equal(s, want)  # the summation order is part of the spec (oracle mo_score_*)
    # batch of codes
    codes = rng.standard_normal((3, D))
    cbk = ops.Codebook(T(E, dev))
    sb = cbk.score(T(codes, dev)).cpu().numpy()
    for b in range(3):
        assert np.array_equal(sb[b], oracle.score_codebook(E, codes[b]))


@pytest.mark.parametrize("K,D,B", [(3001, 512, 64), (1000, 256, 16), (517, 1024, 70), (64, 512, 1), (4096, 512, 33)])
def test_score_batch_mfma(dev, ops, oracle, K, D, B):
    """Batched scoring on the matrix cores: float32 fma chains in the documented order, restated by the oracle
    (numerators bit-identical; the float64 norms differ by an ulp of summation order) and within float32
    rounding of the float64 GEMV path."""
    rng = np.random.default_rng(K + D + B)
    E = rng.standard_normal((K, D)).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    codes = rng.standard_normal((B, D)).astype(np.float32).astype(np.float64)
    codes /= np.linalg.norm(codes, axis=1, keepdims=True)
    codes = codes.astype(np.float32).astype(np.float64)
    cbk = ops.Codebook(T(E, dev))
    sb = cbk.score_batch(T(codes, dev)).cpu().numpy()
    ref = oracle.score_codebook_batch(E, codes)
    np.testing.assert_allclose(sb, ref, rtol=1e-14, atol=0)
    s64 = cbk.score(T(codes, dev)).cpu().numpy()
    assert np.max(np.abs(sb - s64)) < 2e-6


def test_score_golden(dev, ops, golden):
    g = golden("g1_similarity")
    for tag in ("a", "b"):
        cbk = ops.Codebook(T(g[f"{tag}_C"], dev))
        heat = cbk.score(T(g[f"{tag}_q"].astype(np.float64), dev))[0]
        np.testing.assert_allclose(heat.cpu().numpy(), g[f"{tag}_heat"], rtol=0, atol=1e-14)
        x = ops.gather_f64(heat, T(g[f"{tag}_idx"], dev))
        w = ops.softmax_weights(x, True).cpu().numpy()
        np.testing.assert_allclose(w, g[f"{tag}_w_softmax"], rtol=1e-12)
        assert np.max(np.abs(w - g[f"{tag}_w_softmax"])) < 1e-5  # the north-star tolerance
        raw = ops.softmax_weights(x, False).cpu().numpy()
        np.testing.assert_allclose(raw, g[f"{tag}_w_raw"], rtol=0, atol=1e-14)


@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 255, 256, 257, 4095, 4096, 4097, 10000, 100000])
def test_softmax_and_cdf_vs_oracle(dev, ops, oracle, n):
    rng = np.random.default_rng(n)
    x = rng.uniform(-1, 1, n)
    w = ops.softmax_weights(T(x, dev), True).cpu().numpy()
    wref, applied = oracle.softmax_weights(x, True)
    np.testing.assert_allclose(w, wref, rtol=5e-15 if n > 1 else 0)
    # the CDF of identical weights is bit-identical (same blocked order)
    m = rng.uniform(size=n) > 0.3
    if n > 2:
        m[0] = True
    wm = wref * m
    c, status = ops.cdf(T(wm, dev))
    cref, sref = oracle.cdf(wm)
    assert int(status.item()) == sref
    if sref == 0:
        assert np.array_equal(c.cpu().numpy(), cref)


def test_softmax_degenerate_and_nan(dev, ops, oracle):
    x = np.full(1000, 0.25)
    w = ops.softmax_weights(T(x, dev), True).cpu().numpy()
    assert np.array_equal(w, x)  # |max-min| <= 1e-8 -> softmax skipped
    x2 = x.copy()
    x2[10] = np.nan
    w2 = ops.softmax_weights(T(x2, dev), True).cpu().numpy()
    assert np.isnan(w2).all()
    _, st = ops.cdf(T(np.zeros(100), dev))
    assert int(st.item()) == 1
    _, st = ops.cdf(T(np.array([0.1, np.nan, 0.3]), dev))
    assert int(st.item()) == 2


def test_resample_golden_bit_exact(dev, ops, golden):
    from midastouch_amd import _lib
    g = golden("g2_resampler")
    for tag in ["soft4096", "soft1000", "peaky2048", "masked3000", "n1", "n2", "n65"]:
        w = g[f"{tag}_w"]
        c, st = ops.cdf(T(w, dev))
        assert int(st.item()) == 0
        idx = ops.resample_search(c, len(w), _lib.RESAMPLE_MULTINOMIAL, u=T(g[f"{tag}_weighted_random_u"], dev))
        assert np.array_equal(idx.cpu().numpy(), g[f"{tag}_weighted_random_idx"]), tag
        idx = ops.resample_search(c, len(w), _lib.RESAMPLE_SYSTEMATIC, u32=float(g[f"{tag}_low_var_u"][0]))
        assert np.array_equal(idx.cpu().numpy(), g[f"{tag}_low_var_idx"]), tag


def test_resample_philox_matches_oracle(dev, ops, oracle):
    from midastouch_amd import _lib
    rng = np.random.default_rng(21)
    n = 50000
    w = rng.uniform(size=n) ** 4
    c, _ = ops.cdf(T(w, dev))
    cref, _ = oracle.cdf(w)
    idx = ops.resample_search(c, n, _lib.RESAMPLE_MULTINOMIAL, seed=4000, step=5).cpu().numpy()
    assert np.array_equal(idx, oracle.search_lower(cref, oracle.philox_uniform64(n, 4000, 5)))
    idx = ops.resample_search(c, n, _lib.RESAMPLE_SYSTEMATIC, seed=4000, step=5).cpu().numpy()
    assert np.array_equal(idx, oracle.search_systematic(cref, n, oracle.philox_uniform32(4000, 5)))
    assert np.all(np.diff(idx) >= 0)  # systematic resampling is sorted


def test_gather_rows(dev, ops):
    rng = np.random.default_rng(22)
    idx = rng.integers(0, 1000, 5000).astype(np.int32)
    for shape, dt in [((1000, 4, 4), np.float32), ((1000,), np.float64), ((1000,), np.int64), ((1000,), np.float32),
                      ((1000, 3), np.float32), ((1000, 5), np.uint8), ((1000, 256), np.float64)]:
        src = (rng.standard_normal(shape) * 100).astype(dt)
        out = ops.gather_rows(T(src, dev), T(idx, dev)).cpu().numpy()
        assert np.array_equal(out, src[idx]), (shape, dt)


def test_rmse(dev, ops, oracle, golden):
    g = golden("g6_rmse")
    for tag in ("small", "wide", "one", "same"):
        out = ops.rmse(T(g[f"{tag}_poses"], dev), T(g[f"{tag}_gt"], dev)).cpu().numpy()
        rt, rr = oracle.particle_rmse(g[f"{tag}_poses"], g[f"{tag}_gt"])
        assert out[0] == pytest.approx(rt, rel=1e-12)
        assert out[1] == pytest.approx(rr, rel=1e-5, abs=0.03)
        assert out[0] == pytest.approx(float(g[f"{tag}_rmse_t"]), rel=1e-5, abs=1e-9)
        assert out[1] == pytest.approx(float(g[f"{tag}_rmse_r"]), rel=1e-4, abs=0.03)


def test_check_poses(dev, ops):
    P = _rand_poses(1000, 30)
    P[17, 0, 0] = np.nan
    P[400, :3, :3] = 0
    flag, count = ops.check_poses(T(P, dev))
    assert int(count.item()) == 2
    assert set(np.nonzero(flag.cpu().numpy())[0]) == {17, 400}
