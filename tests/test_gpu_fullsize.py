"""BASELINE configs 3 and 4 at their full sizes on one GPU (N = 1 M particles against the replicated 50k codebook; the
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
    e = np.exp(scores[nn] - 1.0)
    S = oracle.blocked_scan(e)[1]
    np.testing.assert_allclose(w, e / S * mask, rtol=1e-12, atol=0)
    ridx, status = oracle.resample_indices(e * mask, "weighted_random", u=oracle.philox_uniform64(N, seed, step))
    assert status == 0 and int(eng.status.cpu()[1]) == int(mask.sum())
    dev_ridx = eng.ridx.cpu().numpy()
    mism = int((dev_ridx != ridx).sum())
    # numpy's exp and the device's may differ in the last place on a few of 10^5..10^6 values: allow a draw or two to
    # land on the other side of a CDF step, never more
    assert mism <= 2, f"{mism} resample indices differ"
    assert np.array_equal(eng.poses.cpu().numpy(), prop[dev_ridx])
    assert np.array_equal(eng.weights_res.cpu().numpy(), w[dev_ridx])
    assert np.array_equal(eng.hint.cpu().numpy(), nn[dev_ridx])
    return mism


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
    np.testing.assert_allclose(sc, oracle.score_codebook(cb.embeddings, traj.codes[2]), rtol=0, atol=1e-14)


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
