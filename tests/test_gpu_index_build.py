"""The per-entry lists of the hint fast paths (neighbour graph, mesh-vertex lists) are built on the device
(midastouch_amd/csrc/index_build.hip).  The host builder of round 1 (MIDAS_HOST_INDEX=1) is their checker: every byte equal."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def _build(ops, feat, verts, poses, dev, host):
    old = os.environ.get("MIDAS_HOST_INDEX")
    os.environ["MIDAS_HOST_INDEX"] = "1" if host else "0"
    try:
        t0 = time.time()
        t6 = ops.Tree(torch.as_tensor(feat).to(dev))
        t3 = ops.Tree(torch.as_tensor(verts).to(dev))
        t6.attach_mesh(t3, torch.as_tensor(poses).to(dev))
        dt = time.time() - t0
    finally:
        if old is None:
            del os.environ["MIDAS_HOST_INDEX"]
        else:
            os.environ["MIDAS_HOST_INDEX"] = old
    return t6, t3, dt


@pytest.mark.parametrize("K,M", [(1, 5), (7, 3), (300, 40), (513, 256), (700, 257), (5000, 2500), (20000, 10000)])
def test_device_built_lists_equal_host_built(dev, K, M):
    from midastouch_amd import ops
    from midastouch_amd.synthetic import make_codebook, r3_se3_host
    cb = make_codebook("004_sugar_box", K=K, D=64, seed=1200 + K, mesh_points=M)
    poses = cb.poses.copy()
    if K >= 300:  # duplicated entries (equal distances: index order) and rotations near pi (twin entries)
        poses[K // 2: K // 2 + 20] = poses[:20]
        from scipy.spatial.transform import Rotation
        R = Rotation.from_rotvec(np.array([[3.1, 0.05, 0.02], [0.0, 3.12, 0.1], [2.2, 2.2, 0.1]])).as_matrix().astype(np.float32)
        poses[5:8, :3, :3] = R
        poses[K - 8: K - 5, :3, :3] = R
        poses[K - 8: K - 5, :3, 3] = poses[5:8, :3, 3] + 1e-4
    feat = r3_se3_host(poses).astype(np.float32)
    a6, a3, t_dev = _build(ops, feat, cb.mesh_vertices, poses, dev, host=False)
    b6, b3, t_host = _build(ops, feat, cb.mesh_vertices, poses, dev, host=True)
    for what in ("nbrs", "rho_out", "twin", "vlist"):
        x, y = a6.export(what), b6.export(what)
        assert x.shape == y.shape
        if not np.array_equal(x.view(np.uint8), y.view(np.uint8)):
            bad = np.argwhere(x.reshape(K, -1) != y.reshape(K, -1))
            raise AssertionError(f"{what}: {len(bad)} differing elements, first at entry {bad[0][0]}, offset {bad[0][1]}")
    if K >= 300:
        assert (a6.export("twin") >= 0).sum() >= 3  # the near-pi entries found their images
    print(f"K={K}: device build {t_dev:.3f} s, host build {t_host:.3f} s")
