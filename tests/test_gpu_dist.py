"""The HIP backend of the particle-sharded frame: shards of ONE GPU stepped in lock-step must reproduce the
fused single-engine frame of all particles bit for bit (kernels + shard offsets; the collectives themselves
are covered by tests/test_dist_cpu.py under gloo).  Needs an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


class FakeComm:
    def __init__(self, r, w):
        self.rank, self.world = r, w

    def all_gather(self, t):
        raise AssertionError("lock-step test never calls the communicator")

    all_to_all = all_gather


@pytest.mark.parametrize("exchange", ["a2a", "allgather", "a2a_fixed", "a2a_fixed_tight", "peer", "peer_c"])
@pytest.mark.parametrize("shards,mode", [(2, "weighted_random"), (3, "low_var"), (1, "weighted_random")])
def test_sharded_hip_equals_fused_engine(dev, shards, mode, exchange):
    from midastouch_amd.dist import HipShardBackend, ShardedFilterEngine, run_lockstep
    from midastouch_amd.engine import FilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    n_loc, K, D = 4096 * 2, 4000, 256
    N = shards * n_loc
    cb = make_codebook(K=K, D=D, seed=1000)
    traj = make_trajectory(cb, T=12, seed=2000)
    rng = np.random.default_rng(0)
    start = cb.poses[rng.integers(0, K, N)]
    single = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, resample=mode, device=dev)
    single.set_particles(torch.as_tensor(start))
    single.project_to_codebook()
    be = HipShardBackend(cb.poses, cb.embeddings, cb.mesh_vertices, dev)
    tight = exchange == "a2a_fixed_tight"  # segments below the expected row count: the overflow block carries the rest
    engs = [ShardedFilterEngine(num_particles=n_loc, backend=be, comm=FakeComm(r, shards), resample=mode,
                                exchange="a2a_fixed" if tight else exchange) for r in range(shards)]
    if tight:
        for e in engs:
            e.seg_cap = (n_loc // shards) * 7 // 8 // 8 * 8
    if exchange in ("peer", "peer_c"):  # shards of one process: the inboxes are plain pointers
        from midastouch_amd.dist import connect_local_peers
        connect_local_peers(engs, exchange)  # "peer_c": the C-side frame (midas_shard_step), records gathered by run_lockstep
    for r, e in enumerate(engs):
        e.set_particles(torch.as_tensor(start[r * n_loc:(r + 1) * n_loc]))
        e.project_to_codebook()
    for t in range(1, 10):
        od, code, gt = (torch.as_tensor(a[t]).to(dev) for a in (traj.odoms, traj.codes, traj.gt_poses))
        single.step(od, code, gt=gt)
        run_lockstep(engs, [((od, code), {"gt": gt}) for _ in engs])
        cat = lambda name: torch.cat([getattr(e, name) for e in engs]).cpu().numpy()
        assert np.array_equal(cat("nn_idx"), single.nn_idx.cpu().numpy()), t
        assert np.array_equal(cat("weights"), single.weights.cpu().numpy()), t
        assert np.array_equal(cat("ridx"), single.ridx.cpu().numpy()), t
        assert np.array_equal(cat("poses"), single.poses.cpu().numpy()), t
        assert np.array_equal(cat("weights_res"), single.weights_res.cpu().numpy()), t
        assert np.array_equal(cat("hint"), single.hint.cpu().numpy()), t
        for e in engs:
            assert np.array_equal(e.status.cpu().numpy(), single.status.cpu().numpy())
            np.testing.assert_allclose(e.rmse.cpu().numpy(), single.rmse.cpu().numpy(), rtol=1e-12)


def test_codebook_row_sharding_equals_replicated(dev, oracle):
    """BASELINE config 4 shape: embedding rows sharded over the ranks, score slices gathered - same frame bit for bit as the
    replicated engine AND as the ORACLE's frame of all particles (scores, weights and indices are spec arithmetic: exact)."""
    from midastouch_amd.dist import HipShardBackend, ShardedFilterEngine, run_lockstep
    from midastouch_amd.engine import FilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    shards, n_loc, K, D = 2, 4096, 4000, 256
    N = shards * n_loc
    cb = make_codebook(K=K, D=D, seed=1000)
    traj = make_trajectory(cb, T=8, seed=2000)
    start = cb.poses[np.random.default_rng(0).integers(0, K, N)]
    single = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
    single.set_particles(torch.as_tensor(start))
    engs = []
    for r in range(shards):
        be = HipShardBackend(cb.poses, cb.embeddings, cb.mesh_vertices, dev, row_shard=(r, shards))
        assert be.codebook.K == K // shards
        e = ShardedFilterEngine(num_particles=n_loc, backend=be, comm=FakeComm(r, shards))
        e.set_particles(torch.as_tensor(start[r * n_loc:(r + 1) * n_loc]))
        engs.append(e)
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    poses = start
    for t in range(1, 6):
        od, code = torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev)
        single.step(od, code)
        run_lockstep(engs, [((od, code), {}) for _ in engs])
        cat = lambda name: torch.cat([getattr(e, name) for e in engs]).cpu().numpy()
        assert np.array_equal(cat("weights"), single.weights.cpu().numpy()), t
        assert np.array_equal(cat("ridx"), single.ridx.cpu().numpy()), t
        assert np.array_equal(cat("poses"), single.poses.cpu().numpy()), t
        # the oracle's frame of all N particles (device Philox draws keyed by the global slot)
        tn, rot = oracle.philox_noise(N, 4000, t - 1, np.float32(2e-4), np.float32(0.5))
        ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn, rot, u=oracle.philox_uniform64(N, 4000, t - 1))
        assert np.array_equal(cat("nn_idx"), ref["nn_idx"]), t
        assert np.array_equal(engs[0].st.scores.cpu().numpy(), ref["scores"]), t  # the gathered slices == the whole codebook's scores
        assert np.array_equal(cat("weights"), ref["weights"]), t
        assert np.array_equal(cat("ridx"), ref["ridx"]), t
        assert np.array_equal(cat("poses"), ref["poses"]), t
        poses = ref["poses"]


@pytest.mark.parametrize("exchange", ["a2a", "allgather", "peer"])
def test_sharded_host_uniforms_replicated(dev, exchange):
    """Parity mode: the uniforms of ALL slots are given to every shard (both exchange forms)."""
    from midastouch_amd.dist import HipShardBackend, ShardedFilterEngine, run_lockstep
    from midastouch_amd.engine import FilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    shards, n_loc, K, D = 2, 4096, 3000, 128
    N = shards * n_loc
    cb = make_codebook(K=K, D=D, seed=1000)
    traj = make_trajectory(cb, T=6, seed=2000)
    start = cb.poses[np.random.default_rng(3).integers(0, K, N)]
    single = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
    single.set_particles(torch.as_tensor(start))
    be = HipShardBackend(cb.poses, cb.embeddings, cb.mesh_vertices, dev)
    engs = [ShardedFilterEngine(num_particles=n_loc, backend=be, comm=FakeComm(r, shards), exchange=exchange) for r in range(shards)]
    if exchange == "peer":
        from midastouch_amd.dist import connect_local_peers
        connect_local_peers(engs)
    for r, e in enumerate(engs):
        e.set_particles(torch.as_tensor(start[r * n_loc:(r + 1) * n_loc]))
    for t in range(1, 5):
        od, code = torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev)
        torch.manual_seed(77 + t)
        u = torch.rand(N, dtype=torch.float64).to(dev)
        single.step(od, code, u=u)
        run_lockstep(engs, [((od, code), {"u": u}) for _ in engs])
        cat = lambda name: torch.cat([getattr(e, name) for e in engs]).cpu().numpy()
        assert np.array_equal(cat("ridx"), single.ridx.cpu().numpy()), t
        assert np.array_equal(cat("poses"), single.poses.cpu().numpy()), t


@pytest.mark.parametrize("exchange", ["a2a", "allgather", "peer_c", "auto"])
def test_single_rank_process_group_nccl(dev, exchange):
    """world_size 1 through torch.distributed's nccl (= RCCL) backend: the real communicator path.  "peer_c" / "auto": the
    whole frame by one C call on the LIBRARY's own RCCL communicator (midas_comm_create: its id travels over torch's
    group), record all_gather by ncclAllGather, inbox + completion flags; then T frames by one midas_shard_run call."""
    import os
    import torch.distributed as dist
    from midastouch_amd.dist import ShardedFilterEngine
    from midastouch_amd.engine import FilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n, K, D = 4096, 2000, 256
        cb = make_codebook(K=K, D=D, seed=1000)
        traj = make_trajectory(cb, T=6, seed=2000)
        start = cb.poses[np.random.default_rng(0).integers(0, K, n)]
        a = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, n, device=dev)
        b = ShardedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, n, device=dev, exchange=exchange)
        assert b.world == 1
        for e in (a, b):
            e.set_particles(torch.as_tensor(start))
            e.project_to_codebook()
        if exchange in ("peer_c", "auto"):
            assert b.exchange == "peer_c" and b._ccomm is not None, b.peer_error
        for t in range(1, 5):
            od, code = torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev)
            a.step(od, code)
            b.step(od, code)
            assert torch.equal(a.ridx, b.ridx) and torch.equal(a.weights, b.weights) and torch.equal(a.poses, b.poses)
        if b.exchange == "peer_c":
            ods, codes = torch.as_tensor(traj.odoms[1:5]).to(dev), torch.as_tensor(traj.codes[1:5]).to(dev)
            for e in (a, b):
                e.set_particles(torch.as_tensor(start))
                e.project_to_codebook()
                e.step_count = 0
            b.run(ods, codes)
            for t in range(4):
                a.step(ods[t], codes[t])
            assert torch.equal(a.ridx, b.ridx) and torch.equal(a.weights, b.weights) and torch.equal(a.poses, b.poses)
            assert int(b.status[0]) == 0
            b.close()
    finally:
        dist.destroy_process_group()


def test_fixed_capacity_exchange_is_loud_when_rows_are_lost(dev):
    """exchange="a2a_fixed" with all the weight mass on ONE shard (the other shard's particles sit off the surface and are
    pruned): every resampled slot draws its source from that shard, far beyond segment + overflow capacity.  The frame
    cannot be completed in that form - it must say so (it used to keep stale rows silently), and the counted
    form on the same cloud is exact."""
    from midastouch_amd.dist import HipShardBackend, ShardedFilterEngine, run_lockstep
    from midastouch_amd.engine import FilterEngine
    from midastouch_amd._lib import MidasError
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    shards, n_loc, K, D = 2, 4096, 3000, 256
    N = shards * n_loc
    cb = make_codebook(K=K, D=D, seed=1000)
    traj = make_trajectory(cb, T=6, seed=2000)
    rng = np.random.default_rng(4)
    start = cb.poses[rng.integers(0, K, N)].copy()
    start[n_loc:, :3, 3] += 0.05  # shard 1: 5 cm off the surface -> pruned (pen_max = 2 mm), zero weight
    be = HipShardBackend(cb.poses, cb.embeddings, cb.mesh_vertices, dev)
    od, code = torch.as_tensor(traj.odoms[1]).to(dev), torch.as_tensor(traj.codes[1]).to(dev)
    engs = [ShardedFilterEngine(num_particles=n_loc, backend=be, comm=FakeComm(r, shards), exchange="a2a_fixed") for r in range(shards)]
    for r, e in enumerate(engs):
        e.set_particles(torch.as_tensor(start[r * n_loc:(r + 1) * n_loc]))
        e.seg_cap, e.ovf_cap = 1024, 512  # (the defaults - 1.5 n / G + 64 and n / 4 - already fall short from four ranks on)
    run_lockstep(engs, [((od, code), {}) for _ in engs])
    # rank 0 owns every source: it sends n_loc rows to rank 1, far more than segment + overflow block hold
    assert engs[0].seg_cap + engs[0].ovf_cap < n_loc
    with pytest.raises(MidasError, match="a2a_fixed.*lost"):
        for e in engs:
            e.ridx  # reading the particle set of a frame that lost rows
    # the counted form completes the same frame: equal to the fused engine
    single = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
    single.set_particles(torch.as_tensor(start))
    single.step(od, code)
    engs = [ShardedFilterEngine(num_particles=n_loc, backend=be, comm=FakeComm(r, shards), exchange="a2a") for r in range(shards)]
    for r, e in enumerate(engs):
        e.set_particles(torch.as_tensor(start[r * n_loc:(r + 1) * n_loc]))
    run_lockstep(engs, [((od, code), {}) for _ in engs])
    assert np.array_equal(torch.cat([e.ridx for e in engs]).cpu().numpy(), single.ridx.cpu().numpy())
    assert int(single.status[1]) <= n_loc  # only shard 0's particles survived the prune
