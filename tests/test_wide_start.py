"""The start bench.py and the full-size c2 test use - `synthetic.wide_start` - IS `particle_filter.init_filter(gt_0, N)` of the class
surface (reference modules/particle_filter.py:124-145, pinned by fixture G8) under the same seed: same draws (translations first), same
"zyx" Euler composition, same float32 product."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


def test_wide_start_is_init_filter_under_the_same_seed():
    from midastouch_amd.config import load_config
    from midastouch_amd.particle_filter import particle_filter
    from midastouch_amd.synthetic import OBJECT_EXTENTS, make_codebook, make_trajectory, mesh_scale, wide_start
    ext = OBJECT_EXTENTS["004_sugar_box"]
    cb = make_codebook("004_sugar_box", K=500, D=128, seed=1001)
    traj = make_trajectory(cb, T=4, seed=2001)
    # a vertex array whose bounding box is the object's: mesh.scale = the box diagonal (:147-151)
    corners = np.array([[sx * ext[0] / 2, sy * ext[1] / 2, sz * ext[2] / 2] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
    pf = particle_filter(load_config(), corners, 1.0, downsample=1, device="cuda:0")  # (no device work in init_filter)
    assert pf.mesh_diagonal() == pytest.approx(mesh_scale(ext), rel=1e-15)
    gt0 = torch.as_tensor(traj.gt_poses[0])
    for seed, N in ((100, 4096), (7, 33)):
        torch.manual_seed(seed)
        ref = pf.init_filter(gt0, N).poses.numpy()
        got = wide_start(ext, traj.gt_poses[0], N, seed)
        assert got.dtype == np.float32 and got.shape == (N, 4, 4)
        assert np.array_equal(got, ref), (seed, N)
