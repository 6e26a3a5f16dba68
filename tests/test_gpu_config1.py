"""BASELINE.json config 1 at its literal size - 004_sugar_box, N = 1000 particles, K = 5000 codebook entries, D = 256 (the
reference's embedding width, config/tcn/default.yaml:19) - with the reference's own random streams (torch CPU mt19937:
add_noise_to_odom's tn then rot, modules/particle_filter.py:326-335; the resampler's torch.multinomial draws, :245), 30 frames,
both engines against the oracle's loop body: propagated poses, NN indices, prune masks, weights, resample indices and resampled
particles bit for bit.  N = 1000 is inside the one-kernel front's small-set range (from 512 particles) and below one summation
block: the dispatch this size takes differs from the 2000 - 5000 particle cases of the other files.  Needs an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

N, K, D, T = 1000, 5000, 256, 31


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def world():
    from midastouch_amd.synthetic import make_codebook, make_trajectory, mesh_scale
    cb = make_codebook("004_sugar_box", K=K, D=D, seed=1000)
    return cb, make_trajectory(cb, T=T + 1, seed=2000), mesh_scale(cb.extents)


def _start(oracle, cb, traj, scale):
    g = torch.Generator().manual_seed(11)  # init_filter's draws (particle_filter.py:133-141) at noise_ratio 0.05
    tn = torch.normal(0.0, scale / 3.0 * 0.05, size=(N, 3), generator=g).numpy()
    rot = torch.normal(0.0, 60.0 * 0.05, size=(N, 3), generator=g).numpy()
    return oracle.init_filter_compose(traj.gt_poses[0], tn, rot)


@pytest.mark.parametrize("mode", ["weighted_random", "low_var"])
def test_config1_exact_size_eager_engine(dev, oracle, world, mode):
    from midastouch_amd.engine import FilterEngine
    cb, traj, scale = world
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    eng = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, resample=mode, device=dev)
    poses = cb.poses[ofl.SE3_NN_idx(_start(oracle, cb, traj, scale))]  # t = 0: projection onto the codebook (filter/filter.py:159-160)
    eng.set_particles(torch.as_tensor(poses))
    eng.project_to_codebook()
    assert np.array_equal(eng.poses.cpu().numpy(), poses)
    for t in range(1, T):
        torch.manual_seed(3000 + t)
        tn = torch.normal(mean=0.0, std=2e-4, size=(N, 3))
        rot = torch.normal(mean=0.0, std=0.5, size=(N, 3))
        if mode == "weighted_random":
            u, u32 = torch.rand(N, dtype=torch.float64), -1.0
        else:
            u, u32 = None, float(torch.rand(1).item())
        ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn.numpy(), rot.numpy(), u=None if u is None else u.numpy(), mode=mode, u32=u32)
        eng.step(torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev), gt=torch.as_tensor(traj.gt_poses[t]).to(dev),
                 tn=tn.to(dev), rot=rot.to(dev), u=None if u is None else u.to(dev), u32=u32)
        assert np.array_equal(eng.poses_prop.cpu().numpy(), ref["poses_prop"]), f"frame {t}: propagated poses"
        assert np.array_equal(eng.nn_idx.cpu().numpy(), ref["nn_idx"]), f"frame {t}: NN index"
        assert np.array_equal(eng.weights.cpu().numpy(), ref["weights"]), f"frame {t}: weights"
        st = eng.status.cpu().numpy()
        assert st[0] == ref["status"] and st[1] == int(ref["mask"].sum())
        assert np.array_equal(eng.ridx.cpu().numpy(), ref["ridx"]), f"frame {t}: resample indices"
        assert np.array_equal(eng.poses.cpu().numpy(), ref["poses"]), f"frame {t}: resampled poses"
        rt, rr = oracle.particle_rmse(ref["poses_prop"], traj.gt_poses[t])
        rm = eng.rmse.cpu().numpy()
        assert rm[0] == pytest.approx(rt, rel=1e-9) and rm[1] == pytest.approx(rr, rel=1e-4, abs=0.03)
        poses = ref["poses"]
    assert len(np.unique(eng.ridx.cpu().numpy())) < N


def test_config1_exact_size_pipelined_engine(dev, oracle, world):
    """The form the bench times (front with the previous frame's resample folded in + tail): the uniforms of frame t are consumed by
    the next call; read back every frame and, for the last ten frames, once at the end."""
    from midastouch_amd.engine import PipelinedFilterEngine
    cb, traj, scale = world
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
    poses = cb.poses[ofl.SE3_NN_idx(_start(oracle, cb, traj, scale))]
    eng.set_particles(torch.as_tensor(poses))
    eng.project_to_codebook()
    ref = None
    for t in range(1, T):
        torch.manual_seed(3000 + t)
        tn = torch.normal(mean=0.0, std=2e-4, size=(N, 3))
        rot = torch.normal(mean=0.0, std=0.5, size=(N, 3))
        u = torch.rand(N, dtype=torch.float64)
        ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn.numpy(), rot.numpy(), u=u.numpy())
        eng.step(torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev), tn=tn.to(dev), rot=rot.to(dev), u=u.to(dev))
        if t <= 20:
            assert np.array_equal(eng.nn_idx.cpu().numpy(), ref["nn_idx"]), f"frame {t}: NN index"
            assert np.array_equal(eng.ridx.cpu().numpy(), ref["ridx"]), f"frame {t}: resample indices"
            assert np.array_equal(eng.weights.cpu().numpy(), ref["weights"]), f"frame {t}: weights"
        poses = ref["poses"]
    assert np.array_equal(eng.ridx.cpu().numpy(), ref["ridx"])
    assert np.array_equal(eng.poses.cpu().numpy(), poses)
    assert np.array_equal(eng.weights.cpu().numpy(), ref["weights"])
