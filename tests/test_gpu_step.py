"""The fused per-frame step (midas_filter_step through FilterEngine) against the oracle's loop body,
free-running over a synthetic trajectory: both sides start from the same particles and consume the
same random draws; every frame must give bit-identical propagated poses, NN indices, prune masks and
resample indices, and weights within 1e-12 relative (north-star bar: 1e-5).  Needs an MI355X.
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


def _setup(N, K, D, seed=0, obj="004_sugar_box"):
    from midastouch_amd.synthetic import make_codebook, make_trajectory, mesh_scale
    cb = make_codebook(obj, K=K, D=D, seed=1000 + seed)
    traj = make_trajectory(cb, T=24, seed=2000 + seed)
    return cb, traj, mesh_scale(cb.extents)


def _init_particles(oracle, cb, traj, scale, N, seed):
    g = torch.Generator().manual_seed(seed)
    tn = torch.normal(0.0, scale / 3.0 * 0.05, size=(N, 3), generator=g).numpy()
    rot = torch.normal(0.0, 60.0 * 0.05, size=(N, 3), generator=g).numpy()
    return oracle.init_filter_compose(traj.gt_poses[0], tn, rot)


def _compare_step(eng, ref, t, check_rmse=None):
    assert np.array_equal(eng.poses_prop.cpu().numpy(), ref["poses_prop"]), f"frame {t}: propagated poses"
    assert np.array_equal(eng.nn_idx.cpu().numpy(), ref["nn_idx"]), f"frame {t}: NN index"
    w = eng.weights.cpu().numpy()
    assert np.array_equal(w == 0, ref["weights"] == 0), f"frame {t}: prune mask"
    assert np.array_equal(w, ref["weights"]), f"frame {t}: weights (scores, exponential and sums are spec arithmetic: exact)"
    status = eng.status.cpu().numpy()
    assert status[0] == ref["status"]
    assert status[1] == int(ref["mask"].sum())
    assert np.array_equal(eng.ridx.cpu().numpy(), ref["ridx"]), f"frame {t}: resample indices"
    assert np.array_equal(eng.poses.cpu().numpy(), ref["poses"]), f"frame {t}: resampled poses"
    np.testing.assert_allclose(eng.weights_res.cpu().numpy(), ref["weights_res"], rtol=1e-12)
    assert np.array_equal(eng.hint.cpu().numpy(), ref["nn_idx_res"])


@pytest.mark.parametrize("mode", ["weighted_random", "low_var"])
def test_step_parity_host_draws(dev, oracle, mode):
    """Parity mode: torch CPU mt19937 draws in the reference's order (tn, rot, then the uniforms)."""
    from midastouch_amd.engine import FilterEngine
    N, K, D = 2000, 5000, 256
    cb, traj, scale = _setup(N, K, D)
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    eng = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, resample=mode, device=dev)
    poses = _init_particles(oracle, cb, traj, scale, N, 5)
    # t = 0: project onto the codebook (filter/filter.py:159-160)
    idx0 = ofl.SE3_NN_idx(poses)
    poses = cb.poses[idx0]
    eng.set_particles(torch.as_tensor(poses))
    eng.project_to_codebook()
    assert np.array_equal(eng.poses.cpu().numpy(), poses)
    for t in range(1, 20):
        torch.manual_seed(3000 + t)
        tn = torch.normal(mean=0.0, std=2e-4, size=(N, 3))
        rot = torch.normal(mean=0.0, std=0.5, size=(N, 3))
        if mode == "weighted_random":
            u, u32 = torch.rand(N, dtype=torch.float64), -1.0
        else:
            u, u32 = None, float(torch.rand(1).item())
        ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn.numpy(), rot.numpy(),
                       u=None if u is None else u.numpy(), mode=mode, u32=u32)
        eng.step(torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev),
                 gt=torch.as_tensor(traj.gt_poses[t]).to(dev), tn=tn.to(dev), rot=rot.to(dev),
                 u=None if u is None else u.to(dev), u32=u32)
        _compare_step(eng, ref, t)
        rt, rr = oracle.particle_rmse(ref["poses_prop"], traj.gt_poses[t])
        rm = eng.rmse.cpu().numpy()
        assert rm[0] == pytest.approx(rt, rel=1e-9) and rm[1] == pytest.approx(rr, rel=1e-4, abs=0.03)
        poses = ref["poses"]
    assert len(np.unique(eng.ridx.cpu().numpy())) < N  # resampling actually concentrated the particles


def test_step_parity_device_philox(dev, oracle):
    """Device mode: the kernels' Philox streams are restated by the oracle, so the run is still exact."""
    from midastouch_amd.engine import FilterEngine
    N, K, D = 3000, 4000, 512
    cb, traj, scale = _setup(N, K, D, seed=1)
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    eng = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4000, device=dev)
    poses = cb.poses[ofl.SE3_NN_idx(_init_particles(oracle, cb, traj, scale, N, 6))]
    eng.set_particles(torch.as_tensor(poses))
    for t in range(1, 16):
        tn, rot = oracle.philox_noise(N, 4000, t - 1, np.float32(2e-4), np.float32(0.5))
        u = oracle.philox_uniform64(N, 4000, t - 1)
        ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn, rot, u=u)
        eng.step(torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev))
        _compare_step(eng, ref, t)
        poses = ref["poses"]


def test_step_hint_does_not_change_results(dev, oracle):
    from midastouch_amd.engine import FilterEngine
    N, K, D = 1500, 3000, 256
    cb, traj, scale = _setup(N, K, D, seed=2)
    outs = []
    for use_hint in (True, False):
        eng = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
        eng.use_hint = use_hint
        eng.set_particles(torch.as_tensor(_init_particles(oracle, cb, traj, scale, N, 7)))
        eng.project_to_codebook()
        for t in range(1, 10):
            eng.step(torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev))
        outs.append((eng.poses.cpu().numpy(), eng.ridx.cpu().numpy(), eng.weights.cpu().numpy()))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_step_all_drifted_and_guards(dev, oracle):
    """Every particle off the surface -> weights all zero -> resampler keeps the particles (status 1)."""
    from midastouch_amd.engine import FilterEngine
    N, K, D = 512, 1000, 256
    cb, traj, scale = _setup(N, K, D, seed=3)
    eng = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
    poses = _init_particles(oracle, cb, traj, scale, N, 8)
    poses[:, :3, 3] += 1.0
    eng.set_particles(torch.as_tensor(poses))
    eng.step(torch.as_tensor(np.eye(4, dtype=np.float32)).to(dev), torch.as_tensor(traj.codes[1]).to(dev))
    st = eng.status.cpu().numpy()
    assert st[0] == 1 and st[1] == 0
    assert np.array_equal(eng.ridx.cpu().numpy(), np.arange(N))
    assert np.array_equal(eng.poses.cpu().numpy(), eng.poses_prop.cpu().numpy())
    assert float(eng.weights.abs().sum().item()) == 0.0


@pytest.mark.parametrize("mode", ["weighted_random", "low_var"])
def test_step_parity_many_particles_coarse_table(dev, oracle, mode):
    """N above 8192 chunks: the resample search goes through the coarse chunk-end table plus probes of the
    global one (k_tail_b2, cshift > 0); ragged last block; device Philox draws."""
    from midastouch_amd.engine import FilterEngine
    N, K, D = 140_003, 1500, 128
    cb, traj, scale = _setup(N, K, D, seed=4)
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    eng = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4100, resample=mode, device=dev)
    rng = np.random.default_rng(9)
    poses = cb.poses[rng.integers(0, K, N)]
    eng.set_particles(torch.as_tensor(poses))
    for t in range(1, 5):
        tn, rot = oracle.philox_noise(N, 4100, t - 1, np.float32(2e-4), np.float32(0.5))
        if mode == "weighted_random":
            u, u32 = oracle.philox_uniform64(N, 4100, t - 1), None
        else:
            u, u32 = None, oracle.philox_uniform32(4100, t - 1)
        ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn, rot, u=u, mode=mode, u32=u32)
        eng.step(torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev))
        _compare_step(eng, ref, t)
        poses = ref["poses"]


@pytest.mark.parametrize("softmax", [True, False])
def test_step_isclose_guard_and_raw_weights(dev, oracle, softmax):
    """Identical codebook rows -> every particle has the same score -> get_similarity's isclose guard skips the
    softmax (particle_filter.py:460-463) and the raw scores become the weights; softmax=False asks for that."""
    from midastouch_amd.engine import FilterEngine
    N, K, D = 9000, 800, 128
    cb, traj, scale = _setup(N, K, D, seed=5)
    emb = np.abs(cb.embeddings)  # non-negative scores: raw weights stay a valid distribution
    if softmax:
        emb[:] = emb[0]
    ofl = oracle.OracleFilter(cb.poses, emb, cb.mesh_vertices)
    eng = FilterEngine(cb.poses, emb, cb.mesh_vertices, N, seed=4200, softmax=softmax, device=dev)
    rng = np.random.default_rng(10)
    poses = cb.poses[rng.integers(0, K, N)]
    eng.set_particles(torch.as_tensor(poses))
    for t in range(1, 4):
        code = np.abs(traj.codes[t])
        tn, rot = oracle.philox_noise(N, 4200, t - 1, np.float32(2e-4), np.float32(0.5))
        u = oracle.philox_uniform64(N, 4200, t - 1)
        ref = ofl.step(poses, traj.odoms[t], code, tn, rot, u=u, softmax=softmax)
        eng.step(torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(code).to(dev))
        if softmax:
            assert len(np.unique(ref["weights"][ref["weights"] != 0])) == 1  # raw, identical scores
            assert ref["weights"].max() < 1.0 / N * 1e3 or ref["weights"].max() > 1e-3  # not softmax-normalised
        assert ref["status"] == 0
        _compare_step(eng, ref, t)
        poses = ref["poses"]


@pytest.mark.parametrize("engine", ["FilterEngine", "PipelinedFilterEngine"])
@pytest.mark.parametrize("mode", ["weighted_random", "low_var"])
def test_step_negative_raw_weights(dev, oracle, engine, mode):
    """Raw weights of a NEGATIVE cosine: every particle scores the same (identical rows: the isclose guard skips the softmax,
    particle_filter.py:459-468), the tactile code points away from them, so w = x * mask <= 0 and sum(w) < 0 - the reference
    resamples with p = w / sum(w) >= 0 (:238).  A lost filter whose cloud has collapsed onto one entry is in this state; the
    division-free probes of the device's search have to turn their comparison round (resample_search.hpp) - the exact walk
    kept the indices right before, at hundreds of microseconds a frame.  Indices, weights and poses against the oracle, and
    the frame must not take the walk's time."""
    from midastouch_amd import engine as E
    N, K, D = 9000, 800, 128
    cb, traj, scale = _setup(N, K, D, seed=5)
    emb = np.abs(cb.embeddings)
    emb[:] = emb[0]
    ofl = oracle.OracleFilter(cb.poses, emb, cb.mesh_vertices)
    eng = getattr(E, engine)(cb.poses, emb, cb.mesh_vertices, N, seed=4200, resample=mode, device=dev)
    rng = np.random.default_rng(10)
    poses = cb.poses[rng.integers(0, K, N)]
    eng.set_particles(torch.as_tensor(poses))
    code = -np.abs(traj.codes[1])
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for t in range(1, 5):
        tn, rot = oracle.philox_noise(N, 4200, t - 1, np.float32(2e-4), np.float32(0.5))
        if mode == "weighted_random":
            u, u32 = oracle.philox_uniform64(N, 4200, t - 1), None
        else:
            u, u32 = None, oracle.philox_uniform32(4200, t - 1)
        ref = ofl.step(poses, traj.odoms[t], code, tn, rot, u=u, mode=mode, u32=u32)
        assert ref["status"] == 0 and ref["weights"].max() <= 0.0 and ref["weights"].min() < 0.0
        ev[0].record()
        eng.step(torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(code).to(dev))
        eng.flush() if hasattr(eng, "flush") else None
        ev[1].record()
        _compare_step(eng, ref, t)
        if t > 1:  # (the first frame loads kernels)
            assert ev[0].elapsed_time(ev[1]) < 1.0, "the search walked instead of probing"
        poses = ref["poses"]


def test_step_unfused_path_odd_dimension(dev, oracle):
    """D = 96 has no fused-front instantiation: separate scoring and particle-update launches, legacy tail
    (midas_filter_step falls back by itself); same parity bar."""
    from midastouch_amd.engine import FilterEngine
    N, K, D = 5000, 2000, 96
    cb, traj, scale = _setup(N, K, D, seed=6)
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    eng = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4300, device=dev)
    rng = np.random.default_rng(11)
    poses = cb.poses[rng.integers(0, K, N)]
    eng.set_particles(torch.as_tensor(poses))
    for t in range(1, 6):
        tn, rot = oracle.philox_noise(N, 4300, t - 1, np.float32(2e-4), np.float32(0.5))
        u = oracle.philox_uniform64(N, 4300, t - 1)
        ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn, rot, u=u)
        eng.step(torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev))
        _compare_step(eng, ref, t)
        poses = ref["poses"]


def test_step_full_size_properties(dev):
    """BASELINE config 2 sizes (N=100k, K=50k, D=512): size-independent properties."""
    from midastouch_amd.engine import FilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    N, K, D = 100_000, 50_000, 512
    cb = make_codebook(K=K, D=D, seed=1002)
    traj = make_trajectory(cb, T=8, seed=2002)
    eng = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
    rng = np.random.default_rng(0)
    start = cb.poses[rng.integers(0, K, N)]
    eng.set_particles(torch.as_tensor(start))
    eng.project_to_codebook()
    for t in range(1, 6):
        eng.step(torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev))
        w = eng.weights
        st = eng.status.cpu().numpy()
        assert st[0] == 0
        assert 0.0 < float(w.sum().item()) <= 1.0 + 1e-9
        ridx = eng.ridx.cpu().numpy()
        assert ridx.min() >= 0 and ridx.max() < N
        assert float(w[torch.as_tensor(ridx).to(dev).long()].min().item()) > 0.0  # never picks a pruned particle
        # gathered poses are exactly the propagated poses at ridx
        assert torch.equal(eng.poses, eng.poses_prop[torch.as_tensor(ridx).to(dev).long()])
        # NN indices are the true nearest codebook entries: re-check a sample by brute force on the GPU
        feat = __import__("midastouch_amd.ops", fromlist=["ops"]).se3_feature(eng.poses_prop[:512])
        d = torch.cdist(feat.double(), eng.cb_feat.double())
        assert torch.equal(d.argmin(dim=1).int(), eng.nn_idx[:512])


@pytest.mark.parametrize("dense", ["0", "1"])
def test_batch_engine_matches_oracle_per_trajectory(dev, oracle, dense, monkeypatch):
    monkeypatch.setenv("MIDAS_DENSE_SCORES", dense)  # 1: the matrix-core pass over all rows; 0: sparse, per trajectory
    """BASELINE config 5 shape (B trajectories per frame): each trajectory of the batch equals the oracle run with
    the matrix-core scores and the Philox streams keyed by b*N + n - indices exact, weights 1e-12."""
    from midastouch_amd.engine import BatchFilterEngine
    B, N, K, D = 5, 1024, 3000, 256
    cb, traj, scale = _setup(N, K, D, seed=4, obj="cotter-pin")
    from midastouch_amd.synthetic import make_trajectory
    trajs = [make_trajectory(cb, T=8, seed=2100 + b) for b in range(B)]
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    eng = BatchFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, B, N, sig_t=1e-4, sig_r=0.5, seed=4000, device=dev)
    rng = np.random.default_rng(3)
    poses = np.stack([cb.poses[rng.integers(0, K, N)] for _ in range(B)])
    eng.set_particles(torch.as_tensor(poses))
    for t in range(1, 6):
        odoms = torch.as_tensor(np.stack([tr.odoms[t] for tr in trajs])).to(dev)
        codes = torch.as_tensor(np.stack([tr.codes[t] for tr in trajs])).to(dev)
        gts = torch.as_tensor(np.stack([tr.gt_poses[t] for tr in trajs])).to(dev)
        eng.step(odoms, codes, gts)
        # sparse scoring (default): the float64 scores of the single-trajectory step; dense: the matrix-core order
        sc = (np.stack([oracle.score_codebook(cb.embeddings, c) for c in codes.cpu().numpy()]) if eng.sparse_scores
              else oracle.score_codebook_batch(cb.embeddings, codes.cpu().numpy()))
        tn_all, rot_all = oracle.philox_noise(B * N, 4000, t - 1, np.float32(1e-4), np.float32(0.5))
        u_all = oracle.philox_uniform64(B * N, 4000, t - 1)
        for b in range(B):
            sl = slice(b * N, (b + 1) * N)
            ref = ofl.step(poses[b], trajs[b].odoms[t], trajs[b].codes[t], tn_all[sl], rot_all[sl], u=u_all[sl], scores=sc[b])
            assert np.array_equal(eng.nn_idx[b].cpu().numpy(), ref["nn_idx"]), (t, b)
            np.testing.assert_allclose(eng.weights[b].cpu().numpy(), ref["weights"], rtol=1e-12, atol=0)
            assert np.array_equal(eng.ridx[b].cpu().numpy(), ref["ridx"]), (t, b)
            assert np.array_equal(eng.poses[b].cpu().numpy(), ref["poses"]), (t, b)
            st = eng.status[b].cpu().numpy()
            assert st[0] == ref["status"] and st[1] == int(ref["mask"].sum())
            rt, rr = oracle.particle_rmse(ref["poses_prop"], trajs[b].gt_poses[t])
            assert eng.rmse[b, 0].item() == pytest.approx(rt, rel=1e-9)
            poses[b] = ref["poses"]


@pytest.mark.parametrize("mode", ["weighted_random", "low_var"])
@pytest.mark.parametrize("read_every,N", [(1, 9000), (3, 9000), (100, 9000), (3, 11000)])
def test_pipelined_batch_engine_equals_batch_engine(dev, mode, read_every, N):
    """midas_lazy_step_batch (every trajectory's resample folded into the next frame's front kernel, trajectory = grid.y) against
    midas_filter_step_batch, which the test above pins to the oracle: NN indices, propagated poses and rmse every frame, the
    materialised particle set whenever it is read - bit-identical, device draws and host uniforms.  The pipelined batch step runs
    PRESORTED (sources + hint-grouped execution order in front of the front kernel): N = 9000 through the one-kernel form with the
    tables in LDS, N = 11 000 (beyond its 10 240 slots) through the two-kernel form; the rmse is formed in slot order either way."""
    from midastouch_amd.engine import BatchFilterEngine, PipelinedBatchFilterEngine
    from midastouch_amd.synthetic import make_trajectory
    B, K, D = 4, 3000, 256   # three summation blocks per trajectory, ragged
    cb, traj, scale = _setup(N, K, D, seed=6, obj="cotter-pin")
    trajs = [make_trajectory(cb, T=12, seed=2300 + b) for b in range(B)]
    rng = np.random.default_rng(5)
    start = torch.as_tensor(np.stack([cb.poses[rng.integers(0, K, N)] for _ in range(B)]))
    engs = [cls(cb.poses, cb.embeddings, cb.mesh_vertices, B, N, sig_t=1e-4, sig_r=0.5, seed=4100, resample=mode, device=dev)
            for cls in (BatchFilterEngine, PipelinedBatchFilterEngine)]
    for e in engs:
        e.set_particles(start)
        e.project_to_codebook()
    gen = torch.Generator().manual_seed(9)
    for t in range(1, 10):
        odoms = torch.as_tensor(np.stack([tr.odoms[t] for tr in trajs])).to(dev)
        codes = torch.as_tensor(np.stack([tr.codes[t] for tr in trajs])).to(dev)
        gts = torch.as_tensor(np.stack([tr.gt_poses[t] for tr in trajs])).to(dev)
        u = torch.rand((B, N), dtype=torch.float64, generator=gen) if (t % 4 == 0 and mode == "weighted_random") else None
        for e in engs:
            e.step(odoms, codes, gts, u=u)
        a, b = engs
        assert torch.equal(a.nn_idx, b.nn_idx), f"frame {t}: NN index"
        assert torch.equal(a.poses_prop, b.poses_prop), f"frame {t}: propagated poses"
        assert torch.equal(a.rmse, b.rmse), f"frame {t}: rmse"
        if t % read_every == 0:
            for name in ("ridx", "poses", "weights", "weights_res", "hint", "status"):
                assert torch.equal(getattr(a, name), getattr(b, name)), f"frame {t}: {name}"
    for name in ("ridx", "poses", "weights", "weights_res", "hint", "status"):
        assert torch.equal(getattr(engs[0], name), getattr(engs[1], name)), name


@pytest.mark.parametrize("B,N", [(1, 16), (2, 17), (3, 4097), (2, 300)])
def test_pipelined_batch_engine_edge_sizes(dev, B, N):
    """Ragged / smallest sizes of the pipelined batch step (one chunk, a chunk and a slot, a block and a slot), no ground truth."""
    from midastouch_amd.engine import BatchFilterEngine, PipelinedBatchFilterEngine
    from midastouch_amd.synthetic import make_trajectory
    K, D = 900, 128
    cb, traj, scale = _setup(N, K, D, seed=8)
    trajs = [make_trajectory(cb, T=8, seed=2400 + b) for b in range(B)]
    rng = np.random.default_rng(B * 1000 + N)
    start = torch.as_tensor(np.stack([cb.poses[rng.integers(0, K, N)] for _ in range(B)]))
    engs = [cls(cb.poses, cb.embeddings, cb.mesh_vertices, B, N, seed=4200, device=dev) for cls in (BatchFilterEngine, PipelinedBatchFilterEngine)]
    for e in engs:
        e.set_particles(start)
    for t in range(1, 6):
        odoms = torch.as_tensor(np.stack([tr.odoms[t] for tr in trajs])).to(dev)
        codes = torch.as_tensor(np.stack([tr.codes[t] for tr in trajs])).to(dev)
        for e in engs:
            e.step(odoms, codes)
        assert torch.equal(engs[0].nn_idx, engs[1].nn_idx) and torch.equal(engs[0].poses_prop, engs[1].poses_prop), f"frame {t}"
        if t in (2, 5):
            for name in ("ridx", "poses", "weights", "weights_res", "hint", "status"):
                assert torch.equal(getattr(engs[0], name), getattr(engs[1], name)), f"frame {t}: {name}"


@pytest.mark.parametrize("engine_name", ["FilterEngine", "PipelinedFilterEngine"])
def test_prune_screen_threshold_boundary(dev, oracle, engine_name):
    """The prune decides from the float32 screening copy of the vertex lists and goes to the float64 records only for a
    vertex within rounding of the threshold (mesh_screen_check).  Particles are placed at thr * (1 + rel) from a mesh
    vertex for rel across [-5e-3, 5e-3] - sure hits, the ambiguous band (about +-8e-4 at this threshold), sure misses - with a threshold below the vertex
    spacing so that this one vertex decides; the mask must be the oracle's exact float64 predicate, particle by particle."""
    import midastouch_amd.engine as E
    K, D, thr = 1500, 128, 1e-4
    cb, traj, scale = _setup(0, K, D, seed=11)
    V = np.asarray(cb.mesh_vertices, dtype=np.float64)
    rng = np.random.default_rng(77)
    rels = np.concatenate([np.linspace(-5e-3, 5e-3, 1601), np.linspace(-4e-6, 4e-6, 801), [0.0] * 46])
    N = len(rels)
    ks = rng.integers(0, K, N)
    poses = cb.poses[ks].copy()
    t0 = poses[:, :3, 3].astype(np.float64)
    vi = np.array([np.argmin(((V - t) ** 2).sum(1)) for t in t0])
    u = rng.normal(size=(N, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    poses[:, :3, 3] = (V[vi] + u * (thr * (1.0 + rels))[:, None]).astype(np.float32)
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices, pen_max=thr)
    eng = getattr(E, engine_name)(cb.poses, cb.embeddings, cb.mesh_vertices, N, pen_max=thr, device=dev)
    eng.set_particles(torch.as_tensor(poses))
    zeros = np.zeros((N, 3), dtype=np.float32)
    odom = np.eye(4, dtype=np.float32)
    uu = np.linspace(0.01, 0.99, N)
    ref = ofl.step(poses, odom, traj.codes[1], zeros, zeros, u=uu)
    eng.step(torch.as_tensor(odom).to(dev), torch.as_tensor(traj.codes[1]).to(dev), tn=torch.as_tensor(zeros).to(dev),
             rot=torch.as_tensor(zeros).to(dev), u=torch.as_tensor(uu).to(dev))
    valid = eng.weights.cpu().numpy() != 0
    assert np.array_equal(valid, ref["mask"]), f"{int((valid != ref['mask']).sum())} masks differ"
    # the construction really straddles the threshold
    assert 0.2 < ref["mask"].mean() < 0.8
    d = ref["dist"]
    assert (np.abs(d / thr - 1.0) < 1e-6).sum() > 50
