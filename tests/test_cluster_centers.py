"""K9 cluster centres: oracle vs scipy's weighted rotation mean (CPU), kernel vs oracle (GPU)."""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from oracle import oracle as orc


def _clustered(n_per, seed, spread_deg=8.0, flat=None):
    """Poses scattered around len(n_per) centres; labels 0..; `flat` = label whose weights are all equal."""
    rng = np.random.default_rng(seed)
    poses, labels, weights = [], [], []
    for c, n in enumerate(n_per):
        Rc = Rotation.random(random_state=seed * 10 + c)
        tc = rng.uniform(-0.1, 0.1, 3)
        R = (Rc * Rotation.from_rotvec(np.deg2rad(spread_deg) * rng.standard_normal((n, 3)))).as_matrix()
        P = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
        P[:, :3, :3] = R.astype(np.float32)
        P[:, :3, 3] = (tc + 2e-3 * rng.standard_normal((n, 3))).astype(np.float32)
        poses.append(P)
        labels.append(np.full(n, c - 1 if c == 0 else c, dtype=np.int64))  # first cluster carries DBSCAN's noise label -1
        w = rng.uniform(0.0, 1.0, n)
        weights.append(np.full(n, 0.37) if flat == c else w)
    poses, labels, weights = np.concatenate(poses), np.concatenate(labels), np.concatenate(weights)
    perm = rng.permutation(len(labels))
    return poses[perm], (weights / weights.sum())[perm], labels[perm]


def test_oracle_matches_scipy_weighted_mean():
    poses, w, labels = _clustered([300, 500, 211], seed=3, flat=2)
    uniq, centers, stds = orc.cluster_centers(poses, w, labels)
    assert list(uniq) == [-1, 1, 2]
    for i, lab in enumerate(uniq):
        sel = labels == lab
        tw = w[sel].astype(np.float32).astype(np.float64)
        if lab == 2:
            assert abs(np.float32(tw.max() - tw.min())) <= 1e-8
            tw = np.ones_like(tw)
        mean_R = Rotation.from_matrix(poses[sel, :3, :3].astype(np.float64)).mean(weights=tw).as_matrix()  # Markley et al.
        assert np.abs(centers[i, :3, :3] - mean_R).max() < 2e-6
        t = poses[sel, :3, 3].astype(np.float64)
        mt = (t * tw[:, None]).sum(0) / tw.sum()
        assert np.abs(centers[i, :3, 3] - mt).max() < 1e-7
        sd = np.sqrt((((t - centers[i, :3, 3].astype(np.float64)) ** 2) * tw[:, None]).sum(0) / tw.sum())
        np.testing.assert_allclose(stds[i], sd, rtol=1e-5)
        assert np.allclose(centers[i, 3], [0, 0, 0, 1])


@pytest.mark.gpu
@pytest.mark.parametrize("n_per,wdtype", [([300, 500, 211], torch.float64), ([1, 70, 4097, 9000], torch.float32), ([5000], torch.float64)])
def test_kernel_matches_oracle(n_per, wdtype):
    from midastouch_amd import ops
    dev = torch.device("cuda", 0)
    poses, w, labels = _clustered(n_per, seed=len(n_per), flat=1 if len(n_per) > 1 else None)
    if wdtype == torch.float32:
        w = w.astype(np.float32)
    uniq, ref_c, ref_s = orc.cluster_centers(poses, w, labels)
    c, s, cnt = ops.cluster_centers(torch.as_tensor(poses).to(dev), torch.as_tensor(w).to(dev), torch.as_tensor(labels).to(dev),
                                    torch.as_tensor(uniq).to(dev))
    assert cnt.cpu().tolist() == [int((labels == u).sum()) for u in uniq]
    assert np.abs(c.cpu().numpy() - ref_c).max() < 2e-6
    np.testing.assert_allclose(s.cpu().numpy(), ref_s, rtol=2e-4, atol=1e-9)
    # same call twice: bit-identical (fixed summation order, no atomics)
    c2, s2, _ = ops.cluster_centers(torch.as_tensor(poses).to(dev), torch.as_tensor(w).to(dev), torch.as_tensor(labels).to(dev),
                                    torch.as_tensor(uniq).to(dev))
    assert torch.equal(c, c2) and torch.equal(s, s2)


@pytest.mark.gpu
def test_class_surface_and_empty_label():
    from midastouch_amd import ops
    from midastouch_amd.particle_filter import Particles, particle_filter
    dev = torch.device("cuda", 0)
    poses, w, labels = _clustered([400, 800], seed=9)
    parts = Particles(torch.as_tensor(poses).to(dev), torch.as_tensor(w).to(dev), torch.as_tensor(labels).to(dev))
    pf = particle_filter.__new__(particle_filter)
    cp, cs = pf.get_cluster_centers(parts, method="quat_avg")
    uniq, ref_c, ref_s = orc.cluster_centers(poses, w, labels)
    assert cp.shape == (2, 4, 4) and cs.shape == (2, 3) and cp.dtype == torch.float32
    assert np.abs(cp.cpu().numpy() - ref_c).max() < 2e-6
    np.testing.assert_allclose(cs.cpu().numpy(), ref_s, rtol=2e-4, atol=1e-9)
    c, s, cnt = ops.cluster_centers(parts.poses, parts.weights, parts.labels, torch.tensor([-1, 1, 7], device=dev))
    assert cnt.cpu().tolist() == [400, 800, 0]
    assert torch.isnan(c[2]).all() and torch.isnan(s[2]).all() and not torch.isnan(c[:2]).any()


def _se3_log_np(P):
    """theseus SE3.log_map restated with scipy / numpy float64: xi = [V^-1 t, omega] (published formulas; th.SE3 itself is not
    installed here - SURVEY 8(c) lists it among the unpinned third-party pieces)."""
    R, t = P[:, :3, :3].astype(np.float64), P[:, :3, 3].astype(np.float64)
    om = Rotation.from_matrix(R).as_rotvec()
    out = np.zeros((len(P), 6))
    for k in range(len(P)):
        th = np.linalg.norm(om[k])
        W = np.array([[0, -om[k, 2], om[k, 1]], [om[k, 2], 0, -om[k, 0]], [-om[k, 1], om[k, 0], 0]])
        if th > 1e-6:
            Vinv = np.eye(3) - 0.5 * W + (1 - th * np.cos(th / 2) / (2 * np.sin(th / 2))) / th**2 * W @ W
        else:
            Vinv = np.eye(3) - 0.5 * W + W @ W / 12.0
        out[k, :3], out[k, 3:] = Vinv @ t[k], om[k]
    return out


def _se3_exp_np(xi):
    u, om = xi[:3], xi[3:]
    th = np.linalg.norm(om)
    W = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    A, B, C = (np.sin(th) / th, (1 - np.cos(th)) / th**2, (th - np.sin(th)) / th**3) if th > 1e-8 else (1.0, 0.5, 1.0 / 6.0)
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + A * W + B * W @ W
    T[:3, 3] = (np.eye(3) + B * W + C * W @ W) @ u
    return T


@pytest.mark.gpu
@pytest.mark.parametrize("n_per", [[500], [300, 900, 41], [2000, 17, 600, 5, 1200]])
def test_logmap_centres_all_clusters_at_once(n_per):
    """get_cluster_centers(method="logmap") - the reference signature's DEFAULT (modules/particle_filter.py:153-206 with
    pose.log_map_averaged, modules/pose.py:101-109) - against a per-cluster float64 restatement of the reference's loop with
    scipy's rotation logarithm: weighted mean of the SE(3) logarithms mapped back with exp, the cluster's float32 weights
    flattened to 1 where isclose(max - min, 0) (:178-184), the spread around the centre's translation (:195-204).  Also: the
    single-cluster helper gives the same pose, and the result does not depend on the run (fixed reduction order)."""
    from midastouch_amd.particle_filter import Particles, particle_filter
    from midastouch_amd.pose import logmap_average_pose
    dev = torch.device("cuda", 0)
    poses, w, labels = _clustered(n_per, seed=21, flat=1 if len(n_per) > 1 else None)
    parts = Particles(torch.as_tensor(poses).to(dev), torch.as_tensor(w).to(dev), torch.as_tensor(labels).to(dev))
    pf = particle_filter.__new__(particle_filter)
    cp, cs = pf.get_cluster_centers(parts)  # method="logmap" is the default
    assert cp.shape == (len(n_per), 4, 4) and cs.shape == (len(n_per), 3) and cp.dtype == torch.float32
    xi = _se3_log_np(poses)
    w32 = w.astype(np.float32)
    for i, lab in enumerate(np.unique(labels)):
        m = labels == lab
        tw = w32[m].astype(np.float64)
        if np.isclose(np.float32(w32[m].max() - w32[m].min()), 0.0):
            tw = np.ones_like(tw)
        ref = _se3_exp_np((xi[m] * tw[:, None]).sum(0) / tw.sum())
        std = np.sqrt((((poses[m, :3, 3].astype(np.float64) - ref[:3, 3]) ** 2) * tw[:, None]).sum(0) / tw.sum())
        assert np.abs(cp[i].cpu().numpy() - ref).max() < 5e-6, (i, lab)
        np.testing.assert_allclose(cs[i].cpu().numpy(), std, rtol=2e-4, atol=1e-8)
        one = logmap_average_pose(parts.poses[torch.as_tensor(m).to(dev)], torch.as_tensor(tw).to(dev))
        assert np.abs(one.cpu().numpy() - ref).max() < 5e-6
    cp2, cs2 = pf.get_cluster_centers(parts, method="logmap")
    assert torch.equal(cp, cp2) and torch.equal(cs, cs2)
