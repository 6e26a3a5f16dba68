"""BASELINE configs 3, 4 and 5 at their full sizes on one GPU (N = 1 M particles against the replicated 50k codebook; the
500k x 512 codebook with N = 100k): what the oracle can restate at that size is compared exactly - the codebook scores,
and everything downstream of the device's NN / prune decisions (weights, blocked CDF, the Philox draws, resample indices,
gathers) - and the NN / prune decisions themselves on a brute-force sample."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def _check_frame(eng, oracle, cb, code, seed, step, sample=512):
    N = eng.N
    nn = eng.nn_idx.cpu().numpy()
    w = eng.weights.cpu().numpy()
    mask = w != 0  # exp(x - 1) > 0: a zero weight is a pruned particle
    prop = eng.poses_prop.cpu().numpy()
    # NN and prune on a sample, by brute force
    rng = np.random.default_rng(step)
    pick = rng.choice(N, sample, replace=False)
    cb_feat = oracle.R3_SE3(cb.poses)
    assert np.array_equal(oracle.nn6(oracle.R3_SE3(prop[pick]), cb_feat)[0], nn[pick])
    assert np.array_equal(~(oracle.nn3_dist(prop[pick], cb.mesh_vertices) > 0.002), mask[pick])
    # downstream of those decisions: exact
    scores = oracle.score_codebook(cb.embeddings, code)
    # the exponential is a spec function (midas_math.hpp exp_spec == oracle mo_exp): weights and indices are EXACT
    e = oracle.exp_spec(scores[nn], 1.0)
    S = oracle.blocked_scan(e)[1]
    assert np.array_equal(w, e / S * mask)
    u = oracle.philox_uniform64(N, seed, step)
    ridx, status = oracle.resample_indices(e * mask, "weighted_random", u=u)
    assert status == 0 and int(eng.status.cpu()[1]) == int(mask.sum())
    dev_ridx = eng.ridx.cpu().numpy()
    mism = int((dev_ridx != ridx).sum())
    assert mism == 0, f"{mism} resample indices differ"
    assert np.array_equal(eng.poses.cpu().numpy(), prop[dev_ridx])
    assert np.array_equal(eng.weights_res.cpu().numpy(), w[dev_ridx])
    assert np.array_equal(eng.hint.cpu().numpy(), nn[dev_ridx])
    # against a math-library exponential (numpy / libm: within 1 ulp of the spec's) a draw or two of 10^5..10^6 may land
    # on the other side of a CDF step (SURVEY 7 hard part 2): reported, bounded, and the weights stay within 1e-12
    e_lib = np.exp(scores[nn] - 1.0)
    np.testing.assert_allclose(w, e_lib / oracle.blocked_scan(e_lib)[1] * mask, rtol=1e-12, atol=0)
    ridx_lib, _ = oracle.resample_indices(e_lib * mask, "weighted_random", u=u)
    mism_lib = int((dev_ridx != ridx_lib).sum())
    assert mism_lib <= 2, f"{mism_lib} resample indices differ from the libm-exp CDF"
    return mism_lib


def _check_propagate(oracle, prev_poses, prop, odom, N, seed, step, sample=2048):
    """The motion model of the frame on a sample: the oracle's fixed-order compose with the frame's Philox normals, bit for bit."""
    pick = np.random.default_rng(1000 + step).choice(N, sample, replace=False)
    tn, rot = oracle.philox_noise(N, seed, step, np.float32(2e-4), np.float32(0.5))
    assert np.array_equal(oracle.propagate(prev_poses[pick], odom, tn[pick], rot[pick]), prop[pick])


@pytest.mark.parametrize("variant", ["FilterEngine", "PipelinedFilterEngine", "run"])
def test_config2_full_size(dev, oracle, variant):
    """c2 - the HEADLINE config - at its exact size (BASELINE.json configs[1]: 004_sugar_box, N = 100 000 particles, 50 000 x 512
    codebook) from the bench's start (reference init_filter(gt_0, N) projected onto the codebook: modules/particle_filter.py:
    129-145, filter/filter.py:159-160), six frames, every frame against the oracle: motion model and NN / prune decisions on a
    brute-force sample, scores of all K rows, weights, blocked CDF, Philox draws, resample indices and gathers exact.  `run` is
    the path bench.py times: midas_lazy_run (T frames by one C call, prediction lists of the sparse scoring on) - the flushed
    particle set after calls of 1, 2 and 3 frames.  The mismatch counts bench.py's parity_probe reports are asserted here:
    0 against the spec exponential (in _check_frame), <= 2 per frame and <= 3 in all against libm's."""
    from midastouch_amd import engine as E
    from midastouch_amd.synthetic import make_codebook, make_trajectory, wide_start
    N, K, D, seed = 100_000, 50_000, 512, 4000
    cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
    traj = make_trajectory(cb, T=8, seed=2001)
    eng = getattr(E, "PipelinedFilterEngine" if variant == "run" else variant)(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=seed, device=dev)
    assert eng.sparse_scores
    eng.set_particles(torch.as_tensor(wide_start(cb.extents, traj.gt_poses[0], N, 100)))
    eng.project_to_codebook()
    proj = eng.poses.cpu().numpy()
    # the projection itself (filter.py:159-160): every particle sits on its nearest codebook pose
    pick = np.random.default_rng(7).choice(N, 512, replace=False)
    start = wide_start(cb.extents, traj.gt_poses[0], N, 100)
    assert np.array_equal(cb.poses[oracle.nn6(oracle.R3_SE3(start[pick]), oracle.R3_SE3(cb.poses))[0]], proj[pick])
    od, co, gt = (torch.as_tensor(a).to(dev) for a in (traj.odoms, traj.codes, traj.gt_poses))
    total, t = 0, 1
    if variant == "run":
        assert eng._score_list is not None  # prediction lists on
        for chunk in (1, 2, 3):
            prev = eng.poses.cpu().numpy() if chunk == 1 else None  # (reading the set materialises it: flush)
            log = eng.run(od[t:t + chunk], co[t:t + chunk], gt[t:t + chunk])
            t += chunk
            last = t - 1
            if prev is not None:
                _check_propagate(oracle, prev, eng.poses_prop.cpu().numpy(), traj.odoms[last], N, seed, last - 1)
            total += _check_frame(eng, oracle, cb, traj.codes[last], seed, last - 1)
            rt, _ = oracle.particle_rmse(eng.poses_prop.cpu().numpy(), traj.gt_poses[last])
            assert float(log[-1, 0]) == pytest.approx(rt, rel=1e-9) and float(eng.rmse[0]) == pytest.approx(rt, rel=1e-9)
        assert eng.step_count == 6
    else:
        for t in range(1, 7):
            prev = eng.poses.cpu().numpy()
            eng.step(od[t], co[t], gt=gt[t])
            _check_propagate(oracle, prev, eng.poses_prop.cpu().numpy(), traj.odoms[t], N, seed, t - 1)
            total += _check_frame(eng, oracle, cb, traj.codes[t], seed, t - 1)
            rt, _ = oracle.particle_rmse(eng.poses_prop.cpu().numpy(), traj.gt_poses[t])
            assert float(eng.rmse[0]) == pytest.approx(rt, rel=1e-9)
    assert total <= 3
    tele = eng.telemetry.cpu().numpy()
    assert tele[0] >= 0 and tele[1] >= 0


@pytest.mark.parametrize("engine", ["FilterEngine", "PipelinedFilterEngine"])
def test_config3_one_million_particles(dev, oracle, engine):
    """c3 on one GPU: N = 1 M particles, 50k x 512 codebook (what each of 8 GPUs holds replicated)."""
    from midastouch_amd import engine as E
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    N, K, D, seed = 1_000_000, 50_000, 512, 4000
    cb = make_codebook("035_power_drill", K=K, D=D, seed=1003)
    traj = make_trajectory(cb, T=6, seed=2003)
    eng = getattr(E, engine)(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=seed, device=dev)
    rng = np.random.default_rng(0)
    eng.set_particles(torch.as_tensor(cb.poses[rng.integers(0, K, N)]))
    total = 0
    for t in range(1, 4):
        eng.step(torch.as_tensor(traj.odoms[t]), torch.as_tensor(traj.codes[t]), gt=torch.as_tensor(traj.gt_poses[t]))
        total += _check_frame(eng, oracle, cb, traj.codes[t], seed, t - 1)
        rt, _ = oracle.particle_rmse(eng.poses_prop.cpu().numpy(), traj.gt_poses[t])
        assert float(eng.rmse[0]) == pytest.approx(rt, rel=1e-9)
    assert total <= 3


def test_config4_half_million_codebook(dev, oracle):
    """c4 on one GPU: the whole 500k x 512 codebook (1 GB of embeddings, 12 GB of index), N = 100k."""
    from midastouch_amd.engine import FilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    N, K, D, seed = 100_000, 500_000, 512, 4000
    cb = make_codebook("025_mug", K=K, D=D, seed=1004)
    traj = make_trajectory(cb, T=5, seed=2004)
    eng = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=seed, device=dev)
    rng = np.random.default_rng(0)
    eng.set_particles(torch.as_tensor(cb.poses[rng.integers(0, K, N)]))
    for t in range(1, 3):
        eng.step(torch.as_tensor(traj.odoms[t]), torch.as_tensor(traj.codes[t]))
        _check_frame(eng, oracle, cb, traj.codes[t], seed, t - 1)
    # the scores themselves, all 500k of them
    sc = eng.codebook.score(torch.as_tensor(traj.codes[2]).to(dev))[0].cpu().numpy()
    assert np.array_equal(sc, oracle.score_codebook(cb.embeddings, traj.codes[2]))


def test_sharded_engine_vs_oracle(dev, oracle):
    """Two particle shards of one GPU stepped in lock-step (both exchange forms) against the ORACLE's frame of all particles
    (the other sharded tests compare with the fused HIP engine)."""
    from midastouch_amd.dist import HipShardBackend, ShardedFilterEngine, run_lockstep
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    from test_gpu_dist import FakeComm
    G, n, K, D, seed = 2, 4096, 3000, 256, 4000
    cb = make_codebook(K=K, D=D, seed=1000)
    traj = make_trajectory(cb, T=8, seed=2000)
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    rng = np.random.default_rng(1)
    start = cb.poses[rng.integers(0, K, G * n)]
    be = HipShardBackend(cb.poses, cb.embeddings, cb.mesh_vertices, dev)
    for exchange in ("allgather", "a2a", "a2a_fixed"):
        engs = [ShardedFilterEngine(num_particles=n, backend=be, comm=FakeComm(r, G), seed=seed, exchange=exchange) for r in range(G)]
        for r, e in enumerate(engs):
            e.set_particles(torch.as_tensor(start[r * n:(r + 1) * n]))
        poses = start
        for t in range(1, 6):
            od, code, gt = (torch.as_tensor(a[t]).to(dev) for a in (traj.odoms, traj.codes, traj.gt_poses))
            run_lockstep(engs, [((od, code), {"gt": gt}) for _ in engs])
            tn, rot = oracle.philox_noise(G * n, seed, t - 1, np.float32(2e-4), np.float32(0.5))
            ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn, rot, u=oracle.philox_uniform64(G * n, seed, t - 1))
            got = lambda name: np.concatenate([getattr(e, name).cpu().numpy() for e in engs])  # noqa: E731
            assert np.array_equal(got("nn_idx"), ref["nn_idx"]), (exchange, t)
            np.testing.assert_allclose(got("weights"), ref["weights"], rtol=1e-12, atol=0)
            assert np.array_equal(got("ridx"), ref["ridx"]), (exchange, t)
            assert np.array_equal(got("poses"), ref["poses"]), (exchange, t)
            poses = ref["poses"]


class _Cat:
    """The shards of one process seen as one engine (what _check_frame reads)."""

    def __init__(self, engs):
        self.engs, self.N = engs, sum(e.N for e in engs)

    def __getattr__(self, name):
        if name == "status":
            return self.engs[0].status
        return torch.cat([getattr(e, name) for e in self.engs])


@pytest.mark.parametrize("exchange", ["peer_c", "a2a"])
def test_config3_partitioned_full_size(dev, oracle, exchange):
    """c3 in its PARTITIONED form at (nearly) full size on one GPU: eight particle shards of 122 880 (30 summation blocks each,
    983 040 particles in all) against the replicated 50k x 512 codebook, stepped in lock-step - the frame of all particles
    (device draws keyed by the global slot) against the oracle's arithmetic downstream of the device's NN / prune decisions,
    which are checked by brute force on a sample; and bit for bit against the fused single-engine frame."""
    from midastouch_amd.dist import HipShardBackend, ShardedFilterEngine, connect_local_peers, run_lockstep
    from midastouch_amd.engine import FilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    from test_gpu_dist import FakeComm
    G, n_loc, K, D, seed = 8, 122_880, 50_000, 512, 4000
    N = G * n_loc
    cb = make_codebook("035_power_drill", K=K, D=D, seed=1003)
    traj = make_trajectory(cb, T=5, seed=2003)
    start = cb.poses[np.random.default_rng(0).integers(0, K, N)]
    be = HipShardBackend(cb.poses, cb.embeddings, cb.mesh_vertices, dev)
    engs = [ShardedFilterEngine(num_particles=n_loc, backend=be, comm=FakeComm(r, G), seed=seed, exchange="a2a") for r in range(G)]
    if exchange == "peer_c":
        connect_local_peers(engs, "peer_c")
    single = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=seed, device=dev)
    single.set_particles(torch.as_tensor(start))
    for r, e in enumerate(engs):
        e.set_particles(torch.as_tensor(start[r * n_loc:(r + 1) * n_loc]))
    whole = _Cat(engs)
    for t in range(1, 3):
        od, code = torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev)
        run_lockstep(engs, [((od, code), {}) for _ in engs])
        single.step(od, code)
        assert torch.equal(whole.nn_idx, single.nn_idx) and torch.equal(whole.ridx, single.ridx), t
        assert torch.equal(whole.poses, single.poses) and torch.equal(whole.weights, single.weights), t
        _check_frame(whole, oracle, cb, traj.codes[t], seed, t - 1)
    for e in engs:
        e.close()


def test_config4_partitioned_full_size(dev, oracle):
    """c4 in its PARTITIONED form at full size on one GPU: the 500k x 512 codebook's embedding rows split over eight shards
    (62 500 rows each; the pose index is shared between the in-process shards, on a node every GPU holds it), eight particle
    shards of 12 288: score slices gathered, then the sharded frame - against the oracle (scores of all 500k rows exact,
    weights / indices exact downstream of the device's NN / prune decisions, those by brute force on a sample)."""
    from midastouch_amd.dist import HipShardBackend, ShardedFilterEngine, run_lockstep
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    from test_gpu_dist import FakeComm
    G, n_loc, K, D, seed = 8, 12_288, 500_000, 512, 4000
    cb = make_codebook("025_mug", K=K, D=D, seed=1004)
    traj = make_trajectory(cb, T=4, seed=2004)
    start = cb.poses[np.random.default_rng(0).integers(0, K, G * n_loc)]
    engs, first = [], None
    for r in range(G):
        be = HipShardBackend(cb.poses, cb.embeddings, cb.mesh_vertices, dev, row_shard=(r, G), share=first)
        first = first or be
        assert be.codebook.K == K // G
        e = ShardedFilterEngine(num_particles=n_loc, backend=be, comm=FakeComm(r, G), seed=seed)
        e.set_particles(torch.as_tensor(start[r * n_loc:(r + 1) * n_loc]))
        engs.append(e)
    whole = _Cat(engs)
    for t in range(1, 3):
        od, code = torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev)
        run_lockstep(engs, [((od, code), {}) for _ in engs])
        _check_frame(whole, oracle, cb, traj.codes[t], seed, t - 1)
        assert np.array_equal(engs[3].st.scores.cpu().numpy(), oracle.score_codebook(cb.embeddings, traj.codes[t])), t


@pytest.mark.parametrize("engine", ["BatchFilterEngine", "PipelinedBatchFilterEngine"])
def test_config5_batch_full_size(dev, oracle, engine):
    """c5 at its full size on one GPU: B = 64 trajectories x N = 10 000 particles against the cotter pin's dense 50k x 512
    codebook (the workload whose certificates need 50 - 100 records: cooperative list scans, float32 prune screen, sparse
    scoring per trajectory).  Three sampled trajectories are followed by the oracle frame by frame with the batch step's
    Philox keys (b N + n): NN indices, prune masks, resample indices and poses exact, weights 1e-12, rmse."""
    from midastouch_amd import engine as E
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    B, N, K, D, seed = 64, 10_000, 50_000, 512, 4200
    cb = make_codebook("cotter-pin", K=K, D=D, seed=1005)
    trajs = [make_trajectory(cb, T=6, seed=2200 + b) for b in range(8)]
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    eng = getattr(E, engine)(cb.poses, cb.embeddings, cb.mesh_vertices, B, N, sig_t=1e-4, sig_r=0.5, seed=seed, device=dev)
    rng = np.random.default_rng(1)
    start = []
    for b in range(B):  # near the truth, as tools/bench_c5.py starts its second run
        d0 = np.linalg.norm(cb.poses[:, :3, 3] - trajs[b % 8].gt_poses[0][:3, 3], axis=1)
        start.append(cb.poses[rng.choice(np.argsort(d0)[:2500], N)])
    start = np.stack(start)
    eng.set_particles(torch.as_tensor(start))
    watch = (0, 37, 63)
    poses = {b: start[b].copy() for b in watch}
    for t in range(1, 5):
        odoms = torch.as_tensor(np.stack([trajs[b % 8].odoms[t] for b in range(B)])).to(dev)
        codes = torch.as_tensor(np.stack([trajs[b % 8].codes[t] for b in range(B)])).to(dev)
        gts = torch.as_tensor(np.stack([trajs[b % 8].gt_poses[t] for b in range(B)])).to(dev)
        eng.step(odoms, codes, gts)
        assert eng.sparse_scores
        tn_all, rot_all = oracle.philox_noise(B * N, seed, t - 1, np.float32(1e-4), np.float32(0.5))
        u_all = oracle.philox_uniform64(B * N, seed, t - 1)
        for b in watch:
            sl = slice(b * N, (b + 1) * N)
            tr = trajs[b % 8]
            ref = ofl.step(poses[b], tr.odoms[t], tr.codes[t], tn_all[sl], rot_all[sl], u=u_all[sl])
            assert np.array_equal(eng.poses_prop[b].cpu().numpy(), ref["poses_prop"]), (t, b)
            assert np.array_equal(eng.nn_idx[b].cpu().numpy(), ref["nn_idx"]), (t, b)
            w = eng.weights[b].cpu().numpy()
            assert np.array_equal(w != 0, ref["mask"]), (t, b)
            np.testing.assert_allclose(w, ref["weights"], rtol=1e-12, atol=0)
            assert np.array_equal(eng.ridx[b].cpu().numpy(), ref["ridx"]), (t, b)
            assert np.array_equal(eng.poses[b].cpu().numpy(), ref["poses"]), (t, b)
            st = eng.status[b].cpu().numpy()
            assert st[0] == ref["status"] and st[1] == int(ref["mask"].sum())
            rt, _ = oracle.particle_rmse(ref["poses_prop"], tr.gt_poses[t])
            assert eng.rmse[b, 0].item() == pytest.approx(rt, rel=1e-9)
            poses[b] = ref["poses"]
    # every trajectory, size-independent properties: resample indices inside the trajectory, kept counts consistent
    ridx = eng.ridx.cpu().numpy()
    assert ridx.shape == (B, N) and ridx.min() >= 0 and ridx.max() < N
    st = eng.status.cpu().numpy()
    w = eng.weights.cpu().numpy()
    assert np.array_equal(st[:, 1], (w != 0).sum(1))
    assert np.isfinite(w).all()


# ---- every particle of a frame by brute force (no sampling) -------------------------------------------------------------
def _brute_force_all(oracle, prop, cb_feat, verts, threads=None):
    """oracle.nn6 / oracle.nn3_dist over ALL rows of `prop`, the rows split across host threads (the C loops release the GIL)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    threads = threads or max(1, min(16, len(os.sched_getaffinity(0))))
    feat = oracle.R3_SE3(prop)
    cuts = np.linspace(0, prop.shape[0], threads + 1).astype(int)
    with ThreadPoolExecutor(threads) as ex:
        nn = list(ex.map(lambda i: oracle.nn6(feat[cuts[i]:cuts[i + 1]], cb_feat)[0], range(threads)))
        d3 = list(ex.map(lambda i: oracle.nn3_dist(prop[cuts[i]:cuts[i + 1]], verts), range(threads)))
    return np.concatenate(nn), np.concatenate(d3)


def test_config2_every_particle_by_brute_force(dev, oracle):
    """c2 at full size, NO sampling: the device's 1-NN (neighbour-list scan with certificate, tree fallback) and prune decisions of
    ALL 100 000 particles against the brute-force scan of the 50 000 entries / of every mesh vertex - in the first frame after
    the wide start (stale hints, the cloud over the whole object: the hardest frame for the certificate) and in a converged one.
    Reference: tactile_tree.SE3_NN (tactile_tree/tactile_tree.py:50-58), remove_invalid_particles (particle_filter.py:386-391)."""
    from midastouch_amd.engine import PipelinedFilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory, wide_start
    N, K, D, seed = 100_000, 50_000, 512, 4000
    cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
    traj = make_trajectory(cb, T=14, seed=2001)
    eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=seed, device=dev)
    eng.set_particles(torch.as_tensor(wide_start(cb.extents, traj.gt_poses[0], N, 100)))
    eng.project_to_codebook()
    cb_feat = oracle.R3_SE3(cb.poses)
    od, co, gt = (torch.as_tensor(a).to(dev) for a in (traj.odoms, traj.codes, traj.gt_poses))
    for t in (1, 12):
        if t > 2:
            eng.run(od[2:t], co[2:t], gt[2:t])
        eng.step(od[t], co[t], gt=gt[t])
        prop = eng.poses_prop.cpu().numpy()
        nn, d3 = _brute_force_all(oracle, prop, cb_feat, cb.mesh_vertices)
        assert np.array_equal(eng.nn_idx.cpu().numpy(), nn), f"frame {t}: {int((eng.nn_idx.cpu().numpy() != nn).sum())} of {N} NN indices differ"
        assert np.array_equal(eng.weights.cpu().numpy() != 0, ~(d3 > 0.002)), f"frame {t}: prune decisions"
    eng.check()


def test_config5_every_particle_by_brute_force(dev, oracle):
    """c5 at full size, NO sampling: NN and prune decisions of all 64 x 10 000 particles of one batch frame by brute force."""
    from midastouch_amd.engine import BatchFilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    B, N, K, D, seed = 64, 10_000, 50_000, 512, 4200
    cb = make_codebook("cotter-pin", K=K, D=D, seed=1005)
    trajs = [make_trajectory(cb, T=4, seed=2200 + b) for b in range(8)]
    eng = BatchFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, B, N, sig_t=1e-4, sig_r=0.5, seed=seed, device=dev)
    rng = np.random.default_rng(1)
    start = []
    for b in range(B):
        d0 = np.linalg.norm(cb.poses[:, :3, 3] - trajs[b % 8].gt_poses[0][:3, 3], axis=1)
        start.append(cb.poses[rng.choice(np.argsort(d0)[:2500], N)])
    eng.set_particles(torch.as_tensor(np.stack(start)))
    for t in (1, 2):
        odoms = torch.as_tensor(np.stack([trajs[b % 8].odoms[t] for b in range(B)])).to(dev)
        codes = torch.as_tensor(np.stack([trajs[b % 8].codes[t] for b in range(B)])).to(dev)
        eng.step(odoms, codes, None)
    prop = eng.poses_prop.cpu().numpy().reshape(B * N, 4, 4)
    nn, d3 = _brute_force_all(oracle, prop, oracle.R3_SE3(cb.poses), cb.mesh_vertices)
    got = eng.nn_idx.cpu().numpy().reshape(-1)
    assert np.array_equal(got, nn), f"{int((got != nn).sum())} of {B * N} NN indices differ"
    assert np.array_equal(eng.weights.cpu().numpy().reshape(-1) != 0, ~(d3 > 0.002))


@pytest.mark.parametrize("engine", ["FilterEngine", "PipelinedFilterEngine"])
def test_prune_exhausted_vertex_list_goes_to_the_tree(dev, oracle, engine):
    """The case the exhaustive c5 check found (two particles of 640 000 pruned that the reference keeps): on the cotter pin's
    dense mesh a particle 2 - 4 mm off its nearest entry walks that entry's whole vertex list (256 records) without a vertex
    within 2 mm and without reaching a record that is provably too far - the list is EXHAUSTED, which decides nothing, and the
    3-d tree has the answer.  Every decision against brute force; the tree search must have been taken."""
    from midastouch_amd import engine as E
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    N, K, D, seed = 30_000, 5_000, 256, 77
    cb = make_codebook("cotter-pin", K=K, D=D, seed=1005)
    traj = make_trajectory(cb, T=3, seed=2200)
    eng = getattr(E, engine)(cb.poses, cb.embeddings, cb.mesh_vertices, N, sig_t=1e-5, sig_r=0.01, seed=seed, device=dev)
    rng = np.random.default_rng(3)
    poses = cb.poses[rng.integers(0, K, N)].copy()
    off = rng.normal(size=(N, 3))
    off *= (rng.uniform(0.0015, 0.0045, N) / np.linalg.norm(off, axis=1))[:, None]
    poses[:, :3, 3] += off.astype(np.float32)
    eng.set_particles(torch.as_tensor(poses))
    tele0 = eng.telemetry.cpu().numpy().copy()
    eye = torch.eye(4)
    eng.step(eye, torch.as_tensor(traj.codes[1]))
    prop = eng.poses_prop.cpu().numpy()
    nn, d3 = _brute_force_all(oracle, prop, oracle.R3_SE3(cb.poses), cb.mesh_vertices)
    assert np.array_equal(eng.nn_idx.cpu().numpy(), nn)
    kept = eng.weights.cpu().numpy() != 0
    assert np.array_equal(kept, ~(d3 > 0.002)), f"{int((kept != ~(d3 > 0.002)).sum())} prune decisions differ"
    assert 0 < kept.sum() < N
    assert int(eng.telemetry.cpu().numpy()[1] - tele0[1]) > 0, "no particle needed the tree: the case is not covered"


@pytest.mark.parametrize("pen_max", [0.0005, 0.004])
def test_prune_thresholds_around_the_fields_reach(dev, oracle, pen_max):
    """The distance field decides "outside the grid = pruned" only for thresholds below the 2.5 mm the grid was grown by; a larger
    `pen.max` (0.004) leaves those particles to the lists / the tree, a small one (0.0005) puts the shell inside the grid: the
    masks are the brute-force masks either way."""
    from midastouch_amd.engine import FilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    N, K, D = 20_000, 4_000, 256
    cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
    traj = make_trajectory(cb, T=3, seed=2001)
    eng = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, sig_t=1e-5, sig_r=0.01, pen_max=pen_max, seed=5, device=dev)
    rng = np.random.default_rng(8)
    poses = cb.poses[rng.integers(0, K, N)].copy()
    off = rng.normal(size=(N, 3))
    off *= (rng.uniform(0.0, 2.5 * pen_max, N) / np.linalg.norm(off, axis=1))[:, None]
    off[: N // 10] *= 20.0  # some far outside the grid
    poses[:, :3, 3] += off.astype(np.float32)
    eng.set_particles(torch.as_tensor(poses))
    eng.step(torch.eye(4), torch.as_tensor(traj.codes[1]))
    prop = eng.poses_prop.cpu().numpy()
    nn, d3 = _brute_force_all(oracle, prop, oracle.R3_SE3(cb.poses), cb.mesh_vertices)
    assert np.array_equal(eng.nn_idx.cpu().numpy(), nn)
    kept = eng.weights.cpu().numpy() != 0
    assert np.array_equal(kept, ~(d3 > pen_max)), f"{int((kept != ~(d3 > pen_max)).sum())} prune decisions differ"
    assert 0 < kept.sum() < N
