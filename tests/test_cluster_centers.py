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
