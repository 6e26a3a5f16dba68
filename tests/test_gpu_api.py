"""The reference-named Python surface (Particles / particle_filter / tactile_tree / particle_rmse) on the
GPU against the golden fixtures of the real reference and against the oracle.  Needs an MI355X."""
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
def setup(dev):
    from midastouch_amd.config import load_config
    from midastouch_amd.particle_filter import particle_filter
    from midastouch_amd.synthetic import make_codebook
    from midastouch_amd.tactile_tree import tactile_tree
    cfg = load_config(["expt.params.num_particles=1500", "expt.codebook_size=3000"])
    cb = make_codebook(K=3000, D=256, seed=1000)
    pf = particle_filter(cfg, cb.mesh_vertices, 1.0, downsample=1, device=dev)
    tree = tactile_tree(torch.as_tensor(cb.poses), torch.as_tensor(cb.cam_poses), torch.as_tensor(cb.embeddings).double())
    tree.to_device(dev)
    return cfg, cb, pf, tree


def test_motion_model_matches_reference_golden(dev, setup, golden):
    """motionModel with the seed the reference used: same draws, poses within float tolerance."""
    from midastouch_amd.particle_filter import Particles
    cfg, cb, pf, tree = setup
    g = golden("g3_motion")
    for tag in ("sim", "mul3"):
        sig_r, sig_t, mul, seed = g[f"{tag}_params"]
        pf.motion_noise = {"mu": 0, "sig_r": float(sig_r), "sig_t": float(sig_t)}
        parts = Particles(torch.as_tensor(g[f"{tag}_poses"]).to(dev))
        torch.manual_seed(int(seed))
        out = pf.motionModel(parts, torch.as_tensor(g[f"{tag}_odom"]).to(dev), multiplier=float(mul))
        np.testing.assert_allclose(out.poses.cpu().numpy(), g[f"{tag}_new_poses"], rtol=0, atol=2e-6)
        assert out is not parts and len(out) == len(parts)
    pf.motion_noise = {"mu": 0, "sig_r": 0.5, "sig_t": 2e-4}


def test_init_filter_matches_reference_golden(dev, setup, golden):
    cfg, cb, pf, tree = setup
    g = golden("g8_init")
    keep = pf.init_noise
    pf.init_noise = [float(v) for v in g["init_noise"]]
    torch.manual_seed(int(g["seed"]))
    parts = pf.init_filter(torch.as_tensor(g["gt"]).to(dev), 1024)
    pf.init_noise = keep
    np.testing.assert_allclose(parts.poses.cpu().numpy(), g["poses"], rtol=0, atol=2e-6)
    assert parts.weights.dtype == torch.float32 and float(parts.weights.sum()) == 1024


def test_reference_goldens_with_every_draw_on_the_device(dev, setup, golden):
    """G3 (motionModel) and G8 (init_filter) - outputs the REFERENCE wrote under torch.manual_seed - with `pf.seed_device_stream(seed)`
    instead of the host generator: the torch.normal draws come from the device replica of torch's stream (mt19937 words + ATen's
    float32 normal transform as tables, torch_normal.py) and give the reference's poses; and the same poses as the host-drawn run,
    exactly."""
    from midastouch_amd.particle_filter import Particles
    cfg, cb, pf, tree = setup
    g = golden("g3_motion")
    try:
        for tag in ("sim", "mul3"):
            sig_r, sig_t, mul, seed = g[f"{tag}_params"]
            pf.motion_noise = {"mu": 0, "sig_r": float(sig_r), "sig_t": float(sig_t)}
            parts = Particles(torch.as_tensor(g[f"{tag}_poses"]).to(dev))
            odom = torch.as_tensor(g[f"{tag}_odom"]).to(dev)
            pf.seed_device_stream(None)
            torch.manual_seed(int(seed))
            host = pf.motionModel(parts, odom, multiplier=float(mul)).poses.cpu().numpy()
            pf.seed_device_stream(int(seed))
            torch.manual_seed(12345)  # (the host generator is not what is drawn from)
            out = pf.motionModel(parts, odom, multiplier=float(mul))
            np.testing.assert_allclose(out.poses.cpu().numpy(), g[f"{tag}_new_poses"], rtol=0, atol=2e-6)
            assert np.array_equal(out.poses.cpu().numpy(), host)
        g8 = golden("g8_init")
        keep = pf.init_noise
        pf.init_noise = [float(v) for v in g8["init_noise"]]
        pf.seed_device_stream(int(g8["seed"]))
        parts = pf.init_filter(torch.as_tensor(g8["gt"]).to(dev), 1024)
        pf.init_noise = keep
        np.testing.assert_allclose(parts.poses.cpu().numpy(), g8["poses"], rtol=0, atol=2e-6)
    finally:
        pf.seed_device_stream(None)
        pf.topk_ties = "index"
        pf.motion_noise = {"mu": 0, "sig_r": 0.5, "sig_t": 2e-4}


def test_get_similarity_matches_reference_golden(dev, setup, golden):
    from midastouch_amd.tactile_tree import tactile_tree
    cfg, cb, pf, _ = setup
    g = golden("g1_similarity")
    for tag in ("a", "b"):
        C, idx, q = g[f"{tag}_C"], g[f"{tag}_idx"], g[f"{tag}_q"]
        qt = torch.as_tensor(q).double()[None].to(dev)
        T = torch.as_tensor(C).double()[idx].to(dev)
        w = pf.get_similarity(qt, T, softmax=True)  # plain (N, D) tensor, as the reference passes
        assert w.dtype == torch.float64
        np.testing.assert_allclose(w.cpu().numpy(), g[f"{tag}_w_softmax"], rtol=1e-12)
        assert np.max(np.abs(w.cpu().numpy() - g[f"{tag}_w_softmax"])) < 1e-5
        np.testing.assert_allclose(pf.get_similarity(qt, T, softmax=False).cpu().numpy(), g[f"{tag}_w_raw"], atol=1e-14)
        # the NNCodes view and the heat-map view give the same numbers
        K = C.shape[0]
        poses = torch.eye(4)[None].repeat(K, 1, 1)
        tr = tactile_tree(poses, poses, torch.as_tensor(C).double())
        tr.to_device(dev)
        from midastouch_amd.tactile_tree import NNCodes
        w2 = pf.get_similarity(qt, NNCodes(tr, torch.as_tensor(idx).to(dev)), softmax=True)
        np.testing.assert_allclose(w2.cpu().numpy(), g[f"{tag}_w_softmax"], rtol=1e-12)
        heat = pf.get_similarity(qt, tr.get_embeddings(), softmax=False)
        np.testing.assert_allclose(heat.cpu().numpy(), g[f"{tag}_heat"], atol=1e-14)
    deg = pf.get_similarity(torch.as_tensor(g["a_q"]).double()[None].to(dev),
                            torch.as_tensor(g["a_C"]).double()[[3] * 50].to(dev), softmax=True)
    np.testing.assert_allclose(deg.cpu().numpy(), g["deg_w"], atol=1e-14)
    one = pf.get_similarity(torch.as_tensor(g["a_q"]).double()[None].to(dev),
                            torch.as_tensor(g["a_C"]).double()[[5]].to(dev), softmax=True)
    assert one.shape == () and abs(float(one) - float(g["one_w"])) < 1e-14


def test_resampler_matches_reference_golden_bit_exact(dev, setup, golden):
    """pf.resampler under the reference's seed reproduces the reference's indices exactly."""
    from midastouch_amd.particle_filter import Particles
    cfg, cb, pf, tree = setup
    g = golden("g2_resampler")
    for tag in ["soft4096", "soft1000", "peaky2048", "masked3000", "n1", "n2", "n65"]:
        n = len(g[f"{tag}_w"])
        poses = torch.eye(4)[None].repeat(n, 1, 1).clone()
        poses[:, 0, 3] = torch.arange(n, dtype=torch.float32)
        for mode in ("weighted_random", "low_var"):
            parts = Particles(poses.to(dev), torch.as_tensor(g[f"{tag}_w"]).to(dev), torch.arange(n, dtype=torch.float32).to(dev))
            torch.manual_seed(int(g[f"{tag}_{mode}_seed"]))
            out = pf.resampler(parts, resample=mode)
            idx = out.poses[:, 0, 3].cpu().numpy().astype(np.int64)
            assert np.array_equal(idx, g[f"{tag}_{mode}_idx"]), (tag, mode)
            assert np.array_equal(out.labels.cpu().numpy().astype(np.int64), idx)
            assert np.array_equal(out.weights.cpu().numpy(), g[f"{tag}_w"][idx])  # weights kept, not reset
    # float32 weights, and the guards
    parts = Particles(torch.eye(4)[None].repeat(500, 1, 1).to(dev), torch.as_tensor(g["f32_w"]).to(dev))
    parts.poses[:, 0, 3] = torch.arange(500, dtype=torch.float32, device=dev)
    torch.manual_seed(77)
    out = pf.resampler(parts)
    assert np.array_equal(out.poses[:, 0, 3].cpu().numpy().astype(np.int32), g["f32_weighted_random_idx"])
    for w in (torch.zeros(10, dtype=torch.float64), torch.tensor([0.1, float("nan"), 0.3], dtype=torch.float64)):
        parts = Particles(torch.eye(4)[None].repeat(len(w), 1, 1).to(dev), w.to(dev))
        assert pf.resampler(parts).poses is parts.poses


def test_remove_invalid_particles_in_place(dev, setup, golden):
    from midastouch_amd.config import load_config
    from midastouch_amd.particle_filter import Particles, particle_filter
    g = golden("g4_prune")
    pf = particle_filter(load_config(), g["verts"], 1.0, downsample=1, device=dev)
    for tag in ("near", "far", "thr"):
        pos = g[f"{tag}_pos"]
        P = torch.eye(4)[None].repeat(len(pos), 1, 1).clone()
        P[:, :3, 3] = torch.as_tensor(pos)
        w = torch.as_tensor(g[f"{tag}_w_in"]).to(dev)
        parts = Particles(P.to(dev), w)
        thr = None if tag != "thr" else float(g["thr_thr"])
        out, drifted = pf.remove_invalid_particles(parts, invalid_dist=thr)
        assert out.weights is w  # mutated in place, like the reference (:401)
        assert np.array_equal(w.cpu().numpy(), g[f"{tag}_w_out"])
        assert bool(drifted) == bool(g[f"{tag}_drifted"])


def test_annealing_matches_reference_golden(dev, setup, golden):
    from midastouch_amd.config import load_config
    from midastouch_amd.particle_filter import Particles, particle_filter
    g = golden("g5_anneal")
    for tag in ("shrink", "grow", "floor"):
        pf = particle_filter(load_config(), np.zeros((4, 3)), 1.0, downsample=1, device=dev)
        w0 = g[f"{tag}_w0"]
        n = len(w0)
        poses = torch.eye(4)[None].repeat(n, 1, 1).clone()
        poses[:, 0, 3] = torch.arange(n, dtype=torch.float32)
        parts = Particles(poses.to(dev), torch.as_tensor(w0).to(dev), torch.arange(n, dtype=torch.float32).to(dev))
        for i, v in enumerate(g[f"{tag}_vars"]):
            parts = pf.annealing(parts, torch.tensor(float(v)), floor=int(g[f"{tag}_floor"]))  # float32 scalar, as in the reference loop
            assert np.array_equal(parts.poses[:, 0, 3].cpu().numpy().astype(np.int32), g[f"{tag}_ids_{i}"]), (tag, i)


def test_particle_rmse_and_se3_nn(dev, setup, oracle, golden):
    from midastouch_amd.particle_filter import Particles, particle_rmse
    cfg, cb, pf, tree = setup
    g = golden("g6_rmse")
    rt, rr = particle_rmse(Particles(torch.as_tensor(g["small_poses"]).to(dev)), torch.as_tensor(g["small_gt"]).to(dev))
    assert float(rt) == pytest.approx(float(g["small_rmse_t"]), rel=1e-5)
    assert float(rr) == pytest.approx(float(g["small_rmse_r"]), rel=1e-4, abs=0.03)
    rng = np.random.default_rng(3)
    q = cb.poses[rng.integers(0, cb.K, 800)].copy()
    q[:, :3, 3] += rng.standard_normal((800, 3)).astype(np.float32) * 5e-4
    p, c, codes = tree.SE3_NN(torch.as_tensor(q).to(dev))
    idx = oracle.nn6(oracle.R3_SE3(q), oracle.R3_SE3(cb.poses))[0]
    assert np.array_equal(codes.idx.cpu().numpy(), idx)
    assert np.array_equal(p.cpu().numpy(), cb.poses[idx]) and np.array_equal(c.cpu().numpy(), cb.cam_poses[idx])
    assert codes.shape == (800, 256) and codes.to_tensor().dtype == torch.float64
    assert np.array_equal(codes.to_tensor().cpu().numpy(), cb.embeddings.astype(np.float64)[idx])
    assert len(tree) == cb.K and tree.get_pose(5).shape == (4, 4) and tree.get_embedding(5).dtype == torch.float64
    # nn > 1 (tactile_tree.py:43-58): the nn nearest per query in order, (N, nn, ...) like the reference's fancy indexing
    p3, c3, e3 = tree.SE3_NN(torch.as_tensor(q[:50]).to(dev), nn=3)
    idx3 = oracle.knn6(oracle.R3_SE3(q[:50]), oracle.R3_SE3(cb.poses), 3)[0]
    assert p3.shape == (50, 3, 4, 4) and c3.shape == (50, 3, 4, 4) and e3.shape == (50, 3, 256) and e3.dtype == torch.float64
    assert np.array_equal(p3.cpu().numpy(), cb.poses[idx3]) and np.array_equal(c3.cpu().numpy(), cb.cam_poses[idx3])
    assert np.array_equal(e3.cpu().numpy(), cb.embeddings.astype(np.float64)[idx3])
    assert np.array_equal(idx3[:, 0], idx[:50])
    p1, _, e1 = tree.SE3_NN(torch.as_tensor(q[0]).to(dev), nn=4)  # a single (4,4) query: squeezed like the reference
    assert p1.shape == (4, 4, 4) and e1.shape == (4, 256)


def test_reference_loop_runs_and_tracks(dev):
    """filter(): the reference's loop order end to end on a synthetic sequence; the estimate tracks."""
    from midastouch_amd.config import load_config
    from midastouch_amd.filter import filter as run_filter, synthetic_sequence
    cfg = load_config(["expt.params.num_particles=3000", "expt.codebook_size=4000"])
    seq = synthetic_sequence(cfg, dev, T=40)
    torch.manual_seed(0)
    stats = run_filter(cfg, seq=seq, device=dev)
    for key in ("rmse_t", "rmse_r", "time", "traj_size", "avg_time", "total_time", "cluster_poses", "cluster_stds",
                "obj_name", "tree_size", "noise_ratio", "init_noise", "init_particles", "num_particles", "log_id", "trial_id"):
        assert key in stats
    assert len(stats["rmse_t"]) == 40 and stats["tree_size"] == 4000
    assert np.isfinite(stats["rmse_t"]).all()
    # converged: the particle cloud ends within a few mm of the ground truth
    assert stats["rmse_t"][-1] < 0.02
    assert min(stats["num_particles"]) >= 1000


def test_filter_real_settings_and_stats_file(dev, tmp_path):
    """filter_real(): raw scores as weights, measurement update every other frame, floor 10000; filter_stats.npy (:252)."""
    from midastouch_amd.config import load_config
    from midastouch_amd.filter import filter_real, synthetic_sequence
    cfg = load_config(["expt.params.num_particles=3000", "expt.codebook_size=4000"])
    seq = synthetic_sequence(cfg, dev, T=24)
    torch.manual_seed(0)
    stats = filter_real(cfg, seq=seq, device=dev, update_freq=2, results_path=str(tmp_path))
    assert len(stats["rmse_t"]) == 24 and np.isfinite(stats["rmse_t"]).all()
    # floor=10000 > 3000 particles: abs(N - floor) never limits the removal, N//3 does; the count stays positive
    assert min(stats["num_particles"]) >= 1
    back = np.load(str(tmp_path / "filter_stats.npy"), allow_pickle=True).item()
    assert back["traj_size"] == 24 and len(back["cluster_poses"]) == 24 and not back["cluster_poses"][0].is_cuda
    assert back["rmse_t"] == stats["rmse_t"]


def test_wallclock_pacing_skips_and_repeats_frames(dev, monkeypatch):
    """pace="wallclock": idx = int(frame_rate * total_time) (filter.py:134-135).  With a host clock that charges 2.5 frame
    periods to every iteration the loop looks at frames 0, 2, 5, 7, ... - frame 2 still initialises (prev_idx == 0, as in the
    reference), the odometry of a skipped stretch is composed from the measured poses - and still tracks; with a clock that
    charges 0.4 periods frames repeat."""
    import types
    from midastouch_amd import filter as filt
    from midastouch_amd.config import load_config
    cfg = load_config(["expt.params.num_particles=3000", "expt.codebook_size=4000"])
    seq = filt.synthetic_sequence(cfg, dev, T=48)
    rate = float(cfg.expt.frame_rate)
    for per_iter in (2.5, 0.5):
        clock = {"t": 0.0}

        def fake_time():  # the loop charges (read after the frame's event) - (read at the top of the iteration)
            clock["t"] += per_iter / rate
            return clock["t"]

        monkeypatch.setattr(filt, "time", types.SimpleNamespace(time=fake_time, sleep=lambda s: None))
        stats = filt.filter(cfg, seq=seq, device=dev, pace="wallclock", max_frames=48)
        seen = np.asarray(stats["frame_idx"])
        steps = np.diff(seen)
        assert seen[0] == 0 and seen[-1] >= 48 - 3 and seen.max() < 48
        if per_iter > 1:   # slow host: frames are skipped, two or three at a time
            assert set(steps.tolist()) <= {2, 3} and abs(steps.mean() - per_iter) < 0.15, seen[:12]
        else:              # fast host: every frame is looked at about twice
            assert set(steps.tolist()) <= {0, 1} and abs(len(seen) - 48 / per_iter) <= 3, seen[:12]
        assert len(stats["rmse_t"]) == len(seen) and np.isfinite(stats["rmse_t"]).all()
        assert stats["rmse_t"][-1] <= 1.5 * stats["rmse_t"][0] + 1e-3  # the estimate does not run away over the gaps
    monkeypatch.undo()
