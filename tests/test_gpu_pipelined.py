"""PipelinedFilterEngine (midas_lazy_step / midas_lazy_flush): the resample of frame t runs inside the front kernel of
frame t+1.  Same parity bar as the eager step: NN indices, propagated poses, resample indices and resampled poses
bit-identical to the oracle, weights within 1e-12 relative - whether or not the particle set is materialised
(read) between frames.  Needs an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def _setup(N, K, D, seed):
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    cb = make_codebook("004_sugar_box", K=K, D=D, seed=1000 + seed)
    traj = make_trajectory(cb, T=24, seed=2000 + seed)
    return cb, traj


@pytest.mark.parametrize("mode", ["weighted_random", "low_var"])
@pytest.mark.parametrize("read_every", [1, 4, 100])
def test_pipelined_parity_device_draws(dev, oracle, mode, read_every):
    from midastouch_amd.engine import PipelinedFilterEngine
    N, K, D = 9000, 3000, 256   # three summation blocks, ragged
    cb, traj = _setup(N, K, D, 7)
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4400, resample=mode, device=dev)
    rng = np.random.default_rng(12)
    poses = cb.poses[rng.integers(0, K, N)]
    eng.set_particles(torch.as_tensor(poses))
    prev = None
    for t in range(1, 14):
        tn, rot = oracle.philox_noise(N, 4400, t - 1, np.float32(2e-4), np.float32(0.5))
        if mode == "weighted_random":
            ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn, rot, u=oracle.philox_uniform64(N, 4400, t - 1))
        else:
            ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn, rot, mode="low_var", u32=oracle.philox_uniform32(4400, t - 1))
        folded = eng._pending and not eng._flushed
        eng.step(torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev),
                 gt=torch.as_tensor(traj.gt_poses[t]).to(dev))
        # always there, no materialisation
        assert np.array_equal(eng.poses_prop.cpu().numpy(), ref["poses_prop"]), f"frame {t}: propagated poses"
        assert np.array_equal(eng.nn_idx.cpu().numpy(), ref["nn_idx"]), f"frame {t}: NN index"
        assert not eng._flushed
        if folded:  # the folded resample reports the previous frame's indices
            assert np.array_equal(eng._ridx.cpu().numpy(), prev["ridx"]), f"frame {t}: folded resample indices"
        if t % read_every == 0:
            assert np.array_equal(eng.ridx.cpu().numpy(), ref["ridx"]), f"frame {t}: resample indices"
            assert eng._flushed
            assert np.array_equal(eng.poses.cpu().numpy(), ref["poses"]), f"frame {t}: resampled poses"
            w = eng.weights.cpu().numpy()
            assert np.array_equal(w == 0, ref["weights"] == 0)
            np.testing.assert_allclose(w, ref["weights"], rtol=1e-12, atol=0)
            np.testing.assert_allclose(eng.weights_res.cpu().numpy(), ref["weights_res"], rtol=1e-12)
            assert np.array_equal(eng.hint.cpu().numpy(), ref["nn_idx_res"])
            st = eng.status.cpu().numpy()
            assert st[0] == ref["status"] and st[1] == int(ref["mask"].sum())
            rt, rr = oracle.particle_rmse(ref["poses_prop"], traj.gt_poses[t])
            rm = eng.rmse.cpu().numpy()
            assert rm[0] == pytest.approx(rt, rel=1e-9) and rm[1] == pytest.approx(rr, rel=1e-4, abs=0.03)
        poses, prev = ref["poses"], ref
    assert np.array_equal(eng.poses.cpu().numpy(), poses)  # final materialisation


def test_pipelined_small_set_two_kernel_front(dev, oracle):
    """N <= 512 in the pipelined form still takes the two-kernel front (resample prologue + feature, then four lanes per particle for
    the list scans: launch_frame_front's rule; larger sets run the single kernel since the folded search has its guide tables)."""
    from midastouch_amd.engine import PipelinedFilterEngine
    N, K, D = 384, 3000, 256
    cb, traj = _setup(N, K, D, 17)
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=5100, device=dev)
    poses = cb.poses[np.random.default_rng(21).integers(0, K, N)]
    eng.set_particles(torch.as_tensor(poses))
    for t in range(1, 10):
        tn, rot = oracle.philox_noise(N, 5100, t - 1, np.float32(2e-4), np.float32(0.5))
        ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn, rot, u=oracle.philox_uniform64(N, 5100, t - 1))
        eng.step(torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev))
        assert np.array_equal(eng.poses_prop.cpu().numpy(), ref["poses_prop"]), f"frame {t}"
        assert np.array_equal(eng.nn_idx.cpu().numpy(), ref["nn_idx"]), f"frame {t}"
        if t % 3 == 0:
            assert np.array_equal(eng.ridx.cpu().numpy(), ref["ridx"]), f"frame {t}"
            assert np.array_equal(eng.poses.cpu().numpy(), ref["poses"]), f"frame {t}"
        poses = ref["poses"]
    assert np.array_equal(eng.ridx.cpu().numpy(), ref["ridx"])


def test_pipelined_parity_host_draws(dev, oracle):
    """Parity mode: the reference's host draws; the uniforms of frame t are consumed by the NEXT call."""
    from midastouch_amd.engine import PipelinedFilterEngine
    N, K, D = 5000, 2500, 128
    cb, traj = _setup(N, K, D, 8)
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
    rng = np.random.default_rng(13)
    poses = cb.poses[rng.integers(0, K, N)]
    eng.set_particles(torch.as_tensor(poses))
    for t in range(1, 9):
        torch.manual_seed(3100 + t)
        tn = torch.normal(mean=0.0, std=2e-4, size=(N, 3))
        rot = torch.normal(mean=0.0, std=0.5, size=(N, 3))
        u = torch.rand(N, dtype=torch.float64)
        ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn.numpy(), rot.numpy(), u=u.numpy())
        ud = u.to(dev)
        eng.step(torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev), tn=tn.to(dev), rot=rot.to(dev), u=ud)
        ud.zero_()  # the engine must have kept its own copy
        assert np.array_equal(eng.nn_idx.cpu().numpy(), ref["nn_idx"]), f"frame {t}"
        poses = ref["poses"]
    assert np.array_equal(eng.ridx.cpu().numpy(), ref["ridx"])
    assert np.array_equal(eng.poses.cpu().numpy(), poses)


def test_pipelined_equals_eager_full_size(dev):
    """c2 sizes: twelve frames, pipelined without reads vs eager - identical particles at the end."""
    from midastouch_amd.engine import FilterEngine, PipelinedFilterEngine
    N, K, D = 100_000, 50_000, 512
    cb, traj = _setup(N, K, D, 9)
    rng = np.random.default_rng(14)
    d0 = np.linalg.norm(cb.poses[:, :3, 3] - traj.gt_poses[0][:3, 3], axis=1)
    start = torch.as_tensor(cb.poses[rng.choice(np.argsort(d0)[:2500], N)])
    od, co = torch.as_tensor(traj.odoms).to(dev), torch.as_tensor(traj.codes).to(dev)
    outs = []
    for cls in (FilterEngine, PipelinedFilterEngine):
        eng = cls(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4500, device=dev)
        eng.set_particles(start)
        eng.project_to_codebook()
        for t in range(1, 13):
            eng.step(od[t], co[t])
        outs.append((eng.poses.cpu().numpy(), eng.ridx.cpu().numpy(), eng.weights.cpu().numpy(), eng.status.cpu().numpy()))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("N", [10240, 10241, 65537, 262144, 262145])
def test_front_kernel_form_boundaries(dev, oracle, N):
    """The single front kernel changes form with the particle count: two kernels up to 10240 particles, one-wave workgroups
    with per-wave resample tables and screened list scans up to 64 summation blocks (262144), workgroup-level tables and
    whole-record scans above.  Either side of both boundaries: pipelined == eager == oracle (indices exact)."""
    from midastouch_amd.engine import FilterEngine, PipelinedFilterEngine
    K, D = 2000, 128
    cb, traj = _setup(N, K, D, 31)
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    rng = np.random.default_rng(N)
    poses = cb.poses[rng.integers(0, K, N)]
    od, co = torch.as_tensor(traj.odoms).to(dev), torch.as_tensor(traj.codes).to(dev)
    engs = [cls(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4700, device=dev) for cls in (FilterEngine, PipelinedFilterEngine)]
    for e in engs:
        e.set_particles(torch.as_tensor(poses))
    for t in range(1, 5):
        for e in engs:
            e.step(od[t], co[t])
        assert torch.equal(engs[0].nn_idx, engs[1].nn_idx) and torch.equal(engs[0].poses_prop, engs[1].poses_prop), f"frame {t}"
        if t <= 2:  # against the oracle (brute-force NN over K entries per particle: a second or two per frame)
            tn, rot = oracle.philox_noise(N, 4700, t - 1, np.float32(2e-4), np.float32(0.5))
            ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn, rot, u=oracle.philox_uniform64(N, 4700, t - 1))
            assert np.array_equal(engs[1].nn_idx.cpu().numpy(), ref["nn_idx"]), f"frame {t}: NN index"
            poses = ref["poses"]
            if t == 2:
                assert np.array_equal(engs[1].ridx.cpu().numpy(), ref["ridx"]), "resample indices"
                assert np.array_equal(engs[1].poses.cpu().numpy(), ref["poses"]), "resampled poses"
    assert torch.equal(engs[0].ridx, engs[1].ridx) and torch.equal(engs[0].poses, engs[1].poses)
    assert torch.equal(engs[0].weights, engs[1].weights)


def test_pipelined_rejects_unsupported_layout(dev):
    from midastouch_amd._lib import MidasError
    from midastouch_amd.engine import PipelinedFilterEngine
    cb, traj = _setup(512, 600, 96, 10)
    with pytest.raises(MidasError):
        PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, 512, device=dev)


@pytest.mark.parametrize("N", [1, 2, 17, 255, 257, 4095, 4097, 8193])
@pytest.mark.parametrize("cls_name", ["FilterEngine", "PipelinedFilterEngine"])
def test_edge_particle_counts(dev, oracle, N, cls_name):
    """Ragged sizes: below a wave, across chunk / workgroup / summation-block boundaries - both engines."""
    import midastouch_amd.engine as E
    K, D = 700, 128
    cb, traj = _setup(N, K, D, 20)
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    eng = getattr(E, cls_name)(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4600, device=dev)
    rng = np.random.default_rng(N)
    poses = cb.poses[rng.integers(0, K, N)]
    eng.set_particles(torch.as_tensor(poses))
    for t in range(1, 6):
        tn, rot = oracle.philox_noise(N, 4600, t - 1, np.float32(2e-4), np.float32(0.5))
        ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn, rot, u=oracle.philox_uniform64(N, 4600, t - 1))
        eng.step(torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev))
        assert np.array_equal(eng.nn_idx.cpu().numpy(), ref["nn_idx"]), f"frame {t}"
        if t % 2 == 0 or t == 5:
            assert eng.status.cpu().numpy()[0] == ref["status"], f"frame {t}"
            assert np.array_equal(eng.ridx.cpu().numpy(), ref["ridx"]), f"frame {t}"
            assert np.array_equal(eng.poses.cpu().numpy(), ref["poses"]), f"frame {t}"
            np.testing.assert_allclose(eng.weights.cpu().numpy(), ref["weights"], rtol=1e-12, atol=0)
        poses = ref["poses"]


def test_run_equals_stepping(dev):
    """midas_lazy_run (T frames by one call) == T calls of midas_lazy_step: bit-identical particle set, per-frame rmse."""
    from midastouch_amd.engine import PipelinedFilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    N, K, D, T = 5000, 4000, 256, 23
    cb = make_codebook(K=K, D=D, seed=1000)
    traj = make_trajectory(cb, T=T + 6, seed=2000)
    rng = np.random.default_rng(3)
    p0 = torch.as_tensor(cb.poses[rng.integers(0, K, N)])
    odoms, codes, gts = (torch.as_tensor(x).to(dev) for x in (traj.odoms, traj.codes, traj.gt_poses))
    a = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4000, device=dev)
    b = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4000, device=dev)
    a.set_particles(p0)
    b.set_particles(p0)
    rm = []
    for t in range(1, 1 + T + 4):
        a.step(odoms[t], codes[t], gt=gts[t])
        if t in (3, 10):
            rm.append(a.rmse.clone())  # the frame's own statistic: no materialisation needed for it
            a.flush()                  # ... this materialises: the next frame starts unfolded, as run() must handle too
    # b: three frames stepped, a flush, then runs of odd and even length, then single steps again
    for t in (1, 2, 3):
        b.step(odoms[t], codes[t], gt=gts[t])
    assert torch.equal(b.rmse, rm[0])
    b.flush()
    log1 = b.run(odoms[4:11], codes[4:11], gts[4:11])          # 7 frames (odd)
    assert torch.equal(b.rmse, rm[1]) and torch.equal(log1[-1, :2], rm[1])
    assert (log1[1:, 2] > log1[:-1, 2]).all()  # device clock at the end of each frame
    keep1 = log1.clone()
    log2 = b.run(odoms[11:1 + T], codes[11:1 + T], gts[11:1 + T])  # even
    torch.cuda.synchronize()
    assert torch.equal(log1, keep1) and log1.data_ptr() != log2.data_ptr()  # an earlier run's log stays valid (a fresh tensor per call)
    for t in range(1 + T, 1 + T + 4):
        b.step(odoms[t], codes[t], gt=gts[t])
    assert b.step_count == a.step_count
    for name in ("poses", "weights", "weights_res", "ridx", "hint", "rmse"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert torch.isfinite(log2).all() and log2.shape == (T - 10, 3)


PRED_CHANCES = 3  # csrc/midas_internal.hpp MIDAS_PRED_CHANCES: frames a listed row stays on the list without being used


def _check_list(listed, used, history, what):
    """The list a frame's tail leaves: every row the frame used, plus rows one of the PRED_CHANCES frames before it used (a listed
    row that goes unused stays listed that many frames); nothing else, nothing twice.  history: the earlier frames' used rows."""
    ls = set(listed.tolist())
    assert len(ls) == listed.numel(), f"{what}: a row is listed twice"
    us = set(used.tolist())
    assert us <= ls, f"{what}: {len(us - ls)} rows in use are not on the list"
    extra = ls - us
    recent = set()
    for h in history[-PRED_CHANCES:]:
        recent |= set(h.tolist())
    if len(history) >= 1:
        assert extra <= recent, f"{what}: {len(extra - recent)} listed rows were not in use in this or the {PRED_CHANCES} frames before"
    if len(history) > PRED_CHANCES:  # a row last used PRED_CHANCES + 1 frames ago has used up its chances
        old = set(history[-PRED_CHANCES - 1].tolist()) - recent - us
        assert not (old & ls), f"{what}: {len(old & ls)} rows are still listed {PRED_CHANCES + 1} frames after their last use"


@pytest.mark.parametrize("N", [20000, 300000])
def test_prediction_list_same_results(dev, oracle, monkeypatch, N):
    """Sparse scoring with the prediction list (the rows a frame used are scored for the next one by streaming workgroups,
    include/midas_hip.h score_list_dev) against the oracle and against the engine without a list: which wave scores a row is
    all that changes.  N = 20000: per-wave tables, one-wave workgroups; N = 300000: workgroup tables, four-wave workgroups.
    Spread start (every particle its own codebook entry: the regime the list is for), stepped frame by frame and by one
    midas_lazy_run call."""
    from midastouch_amd.engine import PipelinedFilterEngine
    K, D, seed = 6000, 256, 4500
    cb, traj = _setup(N, K, D, 9)
    rng = np.random.default_rng(5)
    start = cb.poses[rng.integers(0, K, N)]
    engs = {}
    for tag, env in (("list", "1"), ("nolist", "0")):
        monkeypatch.setenv("MIDAS_SCORE_LIST", env)
        engs[tag] = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=seed, device=dev)
        engs[tag].set_particles(torch.as_tensor(start))
    assert engs["list"]._score_list is not None and engs["nolist"]._score_list is None
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices) if N <= 20000 else None
    poses = start
    history = []
    od, co = torch.as_tensor(traj.odoms).to(dev), torch.as_tensor(traj.codes).to(dev)
    for t in range(1, 8):
        for e in engs.values():
            e.step(od[t], co[t])
        a, b = engs["list"], engs["nolist"]
        assert np.array_equal(a.nn_idx.cpu().numpy(), b.nn_idx.cpu().numpy()), f"frame {t}"
        used = torch.unique(a.nn_idx).long()
        assert torch.equal(a._scores[used], b._scores[used]), f"frame {t}: scores of the rows in use"
        if t >= 2:  # from the second frame on the list is non-empty and most of the rows in use were on it
            par = (a._epoch >> 1) & 1
            n_listed = int(a._score_list[par ^ 1].item())  # written by this frame's tail for the next frame
            listed = a._score_list[2 + (par ^ 1) * K: 2 + (par ^ 1) * K + n_listed].long()
            _check_list(listed, used, history, f"frame {t}")
        history.append(used)
        if t % 3 == 0 or ofl is not None:
            assert np.array_equal(a.ridx.cpu().numpy(), b.ridx.cpu().numpy()), f"frame {t}"
            assert np.array_equal(a.weights.cpu().numpy(), b.weights.cpu().numpy()), f"frame {t}"
        if ofl is not None:
            tn, rot = oracle.philox_noise(N, seed, t - 1, np.float32(2e-4), np.float32(0.5))
            ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn, rot, u=oracle.philox_uniform64(N, seed, t - 1))
            assert np.array_equal(a.nn_idx.cpu().numpy(), ref["nn_idx"]) and np.array_equal(a.ridx.cpu().numpy(), ref["ridx"])
            assert np.array_equal(a.weights.cpu().numpy(), ref["weights"])
            poses = ref["poses"]
    # the same frames again by ONE call (midas_lazy_run advances the epoch by two per frame itself)
    for e in engs.values():
        e.set_particles(torch.as_tensor(start))
        e.step_count = 0
        e.step(od[1], co[1])
        e.run(od[2:8], co[2:8])
    assert np.array_equal(engs["list"].nn_idx.cpu().numpy(), engs["nolist"].nn_idx.cpu().numpy())
    assert np.array_equal(engs["list"].ridx.cpu().numpy(), engs["nolist"].ridx.cpu().numpy())
    assert np.array_equal(engs["list"].weights.cpu().numpy(), engs["nolist"].weights.cpu().numpy())


def test_epoch_wrap_restarts_stamps_and_lists(dev):
    """advance_epoch: before the 32-bit epoch could wrap, the stamps and the list lengths are zeroed and the count restarts
    (a stale stamp equal to a current epoch would make a row look scored: ADVICE round 2)."""
    from midastouch_amd.engine import EPOCH_LIMIT, PipelinedFilterEngine, advance_epoch
    N, K, D = 20000, 3000, 256
    cb, traj = _setup(N, K, D, 3)
    eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=1, device=dev)
    ref = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=1, device=dev)
    rng = np.random.default_rng(2)
    start = torch.as_tensor(cb.poses[rng.integers(0, K, N)])
    eng.set_particles(start)
    ref.set_particles(start)
    od, co = torch.as_tensor(traj.odoms).to(dev), torch.as_tensor(traj.codes).to(dev)
    eng._epoch = EPOCH_LIMIT - 7  # three frames before the restart
    for t in range(1, 9):
        eng.step(od[t], co[t])
        ref.step(od[t], co[t])
        assert 0 < eng._epoch < EPOCH_LIMIT
        assert np.array_equal(eng.nn_idx.cpu().numpy(), ref.nn_idx.cpu().numpy())
        used = torch.unique(eng.nn_idx).long()
        assert torch.equal(eng._scores[used], ref._scores[used]), f"frame {t}"
    assert eng._epoch < 100  # restarted
    assert np.array_equal(eng.ridx.cpu().numpy(), ref.ridx.cpu().numpy())


@pytest.mark.parametrize("T", [1, 2, 3])
def test_run_whose_last_epoch_is_the_largest_allowed(dev, T):
    """advance_epoch lets a run's LAST epoch reach EPOCH_LIMIT - 2; midas_lazy_run checks the whole run's epochs
    before it enqueues anything (it used to re-check after every frame, the last one included, and returned an error with all T
    frames already in flight).  The run succeeds, the engine's state follows, the next call restarts the epochs."""
    from midastouch_amd.engine import EPOCH_LIMIT, PipelinedFilterEngine
    N, K, D = 6000, 3000, 256
    cb, traj = _setup(N, K, D, 3)
    eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=1, device=dev)
    ref = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=1, device=dev)
    rng = np.random.default_rng(2)
    start = torch.as_tensor(cb.poses[rng.integers(0, K, N)])
    eng.set_particles(start)
    ref.set_particles(start)
    od, co = torch.as_tensor(traj.odoms).to(dev), torch.as_tensor(traj.codes).to(dev)
    eng._epoch = EPOCH_LIMIT - 2 - 2 * T   # the run's epochs: LIMIT - 2 T, ..., LIMIT - 2
    eng.run(od[1:1 + T], co[1:1 + T])
    ref.run(od[1:1 + T], co[1:1 + T])
    assert eng._epoch == EPOCH_LIMIT - 2 and eng.step_count == T
    eng.run(od[1 + T:3 + T], co[1 + T:3 + T])
    ref.run(od[1 + T:3 + T], co[1 + T:3 + T])
    assert eng._epoch == 4  # restarted
    eng.check()
    assert np.array_equal(eng.nn_idx.cpu().numpy(), ref.nn_idx.cpu().numpy())
    assert np.array_equal(eng.ridx.cpu().numpy(), ref.ridx.cpu().numpy())
    # one epoch further is refused BEFORE anything is enqueued: the engine's state is untouched
    eng._epoch = EPOCH_LIMIT - 2
    a = eng._epoch
    from midastouch_amd.engine import advance_epoch
    assert advance_epoch(eng, 1) == 2 and a != eng._epoch  # (the host side restarts by itself ...)
    from midastouch_amd._lib import MidasError
    import midastouch_amd.engine as E
    eng._epoch = EPOCH_LIMIT - 2
    orig = E.EPOCH_LIMIT
    try:
        E.EPOCH_LIMIT = 1 << 31  # ... so the C-side check is reached only with the host's restart disabled
        steps = eng.step_count
        with pytest.raises(MidasError):
            eng.run(od[1:3], co[1:3])
        assert eng.step_count == steps
    finally:
        E.EPOCH_LIMIT = orig


def test_prediction_list_seeded_at_projection(dev, monkeypatch):
    """project_to_codebook (filter/filter.py:159-160) knows every particle's nearest entry: the first frame's rows are listed
    there and then (midas_score_list_seed), so the first frame after a wide start scores them with streaming workgroups too.
    Same particle sets as the engine without a list."""
    from midastouch_amd.engine import PipelinedFilterEngine
    N, K, D = 20000, 6000, 256
    cb, traj = _setup(N, K, D, 11)
    rng = np.random.default_rng(6)
    start = cb.poses[rng.integers(0, K, N)].copy()
    start[:, :3, 3] += (rng.standard_normal((N, 3)) * 1e-3).astype(np.float32)
    engs = {}
    for tag, env in (("list", "1"), ("nolist", "0")):
        monkeypatch.setenv("MIDAS_SCORE_LIST", env)
        e = engs[tag] = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=9, device=dev)
        e.set_particles(torch.as_tensor(start))
        idx = e.project_to_codebook()
    a, b = engs["list"], engs["nolist"]
    par = (a._epoch >> 1) & 1
    assert int(a._score_list[par ^ 1].item()) == torch.unique(idx).numel()  # the list the first frame (epoch + 2) will score
    rows0 = int(a.telemetry[2].item())
    od, co = torch.as_tensor(traj.odoms).to(dev), torch.as_tensor(traj.codes).to(dev)
    for t in range(1, 5):
        a.step(od[t], co[t])
        b.step(od[t], co[t])
        assert torch.equal(a.nn_idx, b.nn_idx), t
        if t == 1:  # most of the first frame's rows came off the seeded list, not from claims by particle waves
            listed, claimed = int(a.telemetry[3].item()), int(a.telemetry[2].item()) - rows0
            assert listed == torch.unique(idx).numel() and claimed < listed // 2, (listed, claimed)
    assert torch.equal(a.ridx, b.ridx) and torch.equal(a.weights, b.weights) and torch.equal(a.poses, b.poses)


@pytest.mark.gpu
def test_dense_switch_of_the_prediction_list_same_results(dev, monkeypatch):
    """MIDAS_DENSE_ROWS=<rows> (off by default; measured without gain, kept as a switch): frames whose prediction list holds
    more rows have the streaming waves score the WHOLE codebook and the particle waves only mark the rows they use - decided
    on the device frame by frame.  Same nearest entries, scores of the rows in use, resample indices and weights as without;
    the list keeps following the rows in use, so a later frame below the threshold goes back to the list."""
    from midastouch_amd.engine import PipelinedFilterEngine
    N, K, D, seed = 20000, 6000, 256, 4600
    cb, traj = _setup(N, K, D, 11)
    start = cb.poses[np.random.default_rng(6).integers(0, K, N)]
    od, co = torch.as_tensor(traj.odoms).to(dev), torch.as_tensor(traj.codes).to(dev)
    engs = {}
    for tag in ("dense", "list"):
        engs[tag] = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=seed, device=dev)
        engs[tag].set_particles(torch.as_tensor(start))
        seeded = engs[tag].project_to_codebook()
    tele0 = int(engs["dense"].telemetry[3].item())
    saw_dense = saw_list = False
    history = [torch.unique(torch.as_tensor(seeded)).long().to(dev)]  # (the projection's rows: seeded onto the first list)
    for t in range(1, 9):
        rows_before = int(engs["dense"].telemetry[3].item())
        monkeypatch.setenv("MIDAS_DENSE_ROWS", "1000" if t <= 4 else "100000")  # four dense frames, then back to the list
        engs["dense"].step(od[t], co[t])
        monkeypatch.delenv("MIDAS_DENSE_ROWS")
        engs["list"].step(od[t], co[t])
        a, b = engs["dense"], engs["list"]
        assert torch.equal(a.nn_idx, b.nn_idx), t
        used = torch.unique(a.nn_idx).long()
        assert torch.equal(a._scores[used], b._scores[used]), t
        assert torch.equal(a.ridx, b.ridx) and torch.equal(a.weights, b.weights), t
        par = (a._epoch >> 1) & 1
        n_listed = int(a._score_list[par ^ 1].item())
        # the next frame's list: the rows in use (+ second chances), whatever the mode
        _check_list(a._score_list[2 + (par ^ 1) * K: 2 + (par ^ 1) * K + n_listed].long(), used, history, f"frame {t}")
        history.append(used)
        scored = int(a.telemetry[3].item()) - rows_before
        saw_dense |= scored == K
        saw_list |= 0 < scored < K
    assert saw_dense and saw_list, (saw_dense, saw_list, int(engs["dense"].telemetry[3].item()) - tele0)


@pytest.mark.parametrize("N,K,D", [(20000, 5003, 512), (300000, 3001, 256), (600, 1030, 512)])
def test_dense_front_scores_every_row(dev, oracle, monkeypatch, N, K, D):
    """MIDAS_DENSE_SCORES=1: the front launch streams ALL K rows beside the particle waves (the K1 GEMV of SURVEY 8(a), two
    quads of rows per scoring wave - score_wave_multi) instead of scoring only the rows in use.  Every one of the K scores equals
    the oracle's bit for bit (K not a multiple of the eight rows a scoring wave takes: the ragged end), and the particle sets are
    those of the sparse engine.  N = 20000: one-wave workgroups with per-wave tables; 300000: four-wave workgroups; 600: the
    two-kernel front (its scoring workgroups keep one quad per wave)."""
    from midastouch_amd.engine import PipelinedFilterEngine
    cb, traj = _setup(N, K, D, 13)
    start = cb.poses[np.random.default_rng(8).integers(0, K, N)]
    engs = {}
    for tag, env in (("dense", "1"), ("sparse", "0")):
        monkeypatch.setenv("MIDAS_DENSE_SCORES", env)
        engs[tag] = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=77, device=dev)
        engs[tag].set_particles(torch.as_tensor(start))
    assert not engs["dense"].sparse_scores and engs["sparse"].sparse_scores
    od, co = torch.as_tensor(traj.odoms).to(dev), torch.as_tensor(traj.codes).to(dev)
    for t in range(1, 6):
        for e in engs.values():
            e.step(od[t], co[t])
        a, b = engs["dense"], engs["sparse"]
        ref = oracle.score_codebook(cb.embeddings, traj.codes[t])
        a.flush()
        assert np.array_equal(a._scores.cpu().numpy(), ref), f"frame {t}: {int((a._scores.cpu().numpy() != ref).sum())} of {K} scores differ"
        assert torch.equal(a.nn_idx, b.nn_idx) and torch.equal(a.ridx, b.ridx) and torch.equal(a.weights, b.weights), f"frame {t}"
    for e in engs.values():
        e.run(od[6:12], co[6:12])
    assert torch.equal(engs["dense"].ridx, engs["sparse"].ridx) and torch.equal(engs["dense"].poses, engs["sparse"].poses)
    assert np.array_equal(engs["dense"]._scores.cpu().numpy(), oracle.score_codebook(cb.embeddings, traj.codes[11]))


def test_pipelined_every_draw_from_the_device_replica_of_torchs_stream(dev, oracle):
    """seed_torch_stream(seed, motion=True): motion noise AND resample draws from the device replica of torch's CPU generator, in
    the reference's order (tn, rot, then the resampler's uniforms - particle_filter.py:326-335, :245), unit normals drawn a frame
    ahead: frame after frame identical to the oracle fed with the host generator's own draws under torch.manual_seed."""
    from midastouch_amd.engine import PipelinedFilterEngine
    N, K, D = 20_000, 3000, 256
    cb, traj = _setup(N, K, D, 7)
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=1, device=dev)
    rng = np.random.default_rng(12)
    poses = cb.poses[rng.integers(0, K, N)]
    eng.set_particles(torch.as_tensor(poses))
    eng.seed_torch_stream(2024, motion=True)
    torch.manual_seed(2024)
    for t in range(1, 9):
        tn = torch.normal(0.0, 2e-4, size=(N, 3)).numpy()
        rot = torch.normal(0.0, 0.5, size=(N, 3)).numpy()
        u = torch.rand(N, dtype=torch.float64).numpy()
        ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn, rot, u=u)
        eng.step(torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev))
        assert np.array_equal(eng.poses_prop.cpu().numpy(), ref["poses_prop"]), f"frame {t}: propagated poses"
        assert np.array_equal(eng.nn_idx.cpu().numpy(), ref["nn_idx"]), f"frame {t}"
        if t % 3 == 0:
            assert np.array_equal(eng.ridx.cpu().numpy(), ref["ridx"]), f"frame {t}: resample indices"
            assert np.array_equal(eng.poses.cpu().numpy(), ref["poses"]), f"frame {t}"
        poses = ref["poses"]
