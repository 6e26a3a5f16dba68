"""The particle-sharded frame (midastouch_amd/dist.py) under torch.distributed gloo, world_size 2, on CPU.

The per-shard kernels are replaced by an oracle-backed backend (tests/_oracle_shard_backend.py) - the
thing under test here is the sharding logic and its collectives: offsets, gathered partials, global CDF,
cross-rank resample.  The 2-rank run must reproduce the single-process oracle run of all particles
bit for bit (N per rank is a multiple of the 4096-slot summation block).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

N_LOC, K, D, FRAMES = 4096, 1500, 64, 4


def _data():
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    cb = make_codebook(K=K, D=D, seed=1000)
    traj = make_trajectory(cb, T=FRAMES + 1, seed=2000)
    rng = np.random.default_rng(0)
    start = cb.poses[rng.integers(0, K, 2 * N_LOC)]
    return cb, traj, start


def _worker(rank, world, port, mode, exchange, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from midastouch_amd.dist import ShardedFilterEngine
    from tests._oracle_shard_backend import OracleShardBackend
    cb, traj, start = _data()
    tight = exchange == "a2a_fixed_tight"  # segments smaller than the expected row count: the overflow block is in use
    eng = ShardedFilterEngine(num_particles=N_LOC, backend=OracleShardBackend(cb.poses, cb.embeddings, cb.mesh_vertices),
                              resample=mode, seed=4000, exchange="a2a_fixed" if tight else exchange)
    if tight:
        eng.seg_cap = (N_LOC // world) * 7 // 8 // 8 * 8
    assert eng.world == world and eng.rank == rank
    eng.set_particles(torch.as_tensor(start[rank * N_LOC:(rank + 1) * N_LOC]))
    res = []
    for t in range(1, FRAMES + 1):
        eng.step(torch.as_tensor(traj.odoms[t]), torch.as_tensor(traj.codes[t]), gt=torch.as_tensor(traj.gt_poses[t]))
        res.append({"ridx": eng.ridx.numpy().copy(), "weights": eng.weights.numpy().copy(),
                    "poses": eng.poses.numpy().copy(), "status": eng.status.numpy().copy(), "rmse": eng.rmse.numpy().copy()})
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("exchange", ["a2a", "allgather", "a2a_fixed", "a2a_fixed_tight"])
@pytest.mark.parametrize("mode", ["weighted_random", "low_var"])
def test_two_rank_gloo_equals_single_process(tmp_path, oracle, mode, exchange):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, exchange, str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=False) for r in range(world)]
    cb, traj, start = _data()
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    poses = start
    N = world * N_LOC
    for t in range(1, FRAMES + 1):
        tn, rot = oracle.philox_noise(N, 4000, t - 1, np.float32(2e-4), np.float32(0.5))
        if mode == "weighted_random":
            ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn, rot, u=oracle.philox_uniform64(N, 4000, t - 1))
        else:
            ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn, rot, mode="low_var", u32=oracle.philox_uniform32(4000, t - 1))
        ridx = np.concatenate([p[t - 1]["ridx"] for p in parts])
        w = np.concatenate([p[t - 1]["weights"] for p in parts])
        ps = np.concatenate([p[t - 1]["poses"] for p in parts])
        assert np.array_equal(ridx, ref["ridx"]), f"frame {t}"
        assert np.array_equal(w, ref["weights"]), f"frame {t}"
        assert np.array_equal(ps, ref["poses"]), f"frame {t}"
        for p in parts:
            assert p[t - 1]["status"][0] == ref["status"] and p[t - 1]["status"][1] == int(ref["mask"].sum())
        rt, rr = oracle.particle_rmse(ref["poses_prop"], traj.gt_poses[t])
        assert parts[0][t - 1]["rmse"][0] == pytest.approx(rt, rel=1e-9)
        assert np.array_equal(parts[0][t - 1]["rmse"], parts[1][t - 1]["rmse"])
        poses = ref["poses"]


def test_four_rank_gloo_auto_exchange(tmp_path, oracle):
    """Four ranks: exchange='auto' picks the owner-side all_to_all without read-back; same bar (bit-identical to one process)."""
    global N_LOC
    world = 4
    mp.spawn(_worker4, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=False) for r in range(world)]
    assert all(p["exchange"] == "a2a_fixed" for p in parts)
    cb, traj, _ = _data()
    start = cb.poses[np.random.default_rng(5).integers(0, K, world * N_LOC)]
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    poses, N = start, world * N_LOC
    for t in range(1, 3):
        tn, rot = oracle.philox_noise(N, 4000, t - 1, np.float32(2e-4), np.float32(0.5))
        ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn, rot, u=oracle.philox_uniform64(N, 4000, t - 1))
        assert np.array_equal(np.concatenate([p["res"][t - 1]["ridx"] for p in parts]), ref["ridx"]), f"frame {t}"
        assert np.array_equal(np.concatenate([p["res"][t - 1]["poses"] for p in parts]), ref["poses"]), f"frame {t}"
        assert np.array_equal(np.concatenate([p["res"][t - 1]["weights"] for p in parts]), ref["weights"]), f"frame {t}"
        poses = ref["poses"]


def _worker4(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from midastouch_amd.dist import ShardedFilterEngine
    from tests._oracle_shard_backend import OracleShardBackend
    cb, traj, _ = _data()
    start = cb.poses[np.random.default_rng(5).integers(0, K, world * N_LOC)]
    eng = ShardedFilterEngine(num_particles=N_LOC, backend=OracleShardBackend(cb.poses, cb.embeddings, cb.mesh_vertices), seed=4000)
    eng.set_particles(torch.as_tensor(start[rank * N_LOC:(rank + 1) * N_LOC]))
    res = []
    for t in range(1, 3):
        eng.step(torch.as_tensor(traj.odoms[t]), torch.as_tensor(traj.codes[t]))
        res.append({"ridx": eng.ridx.numpy().copy(), "weights": eng.weights.numpy().copy(), "poses": eng.poses.numpy().copy()})
    torch.save({"exchange": eng.exchange, "res": res}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_lockstep_three_shards_equal_single(oracle):
    """Same check without any collective library: three shards of one process stepped in lock-step."""
    from midastouch_amd.dist import ShardedFilterEngine, run_lockstep
    from tests._oracle_shard_backend import OracleShardBackend
    cb, traj, start = _data()
    rng = np.random.default_rng(1)
    start = cb.poses[rng.integers(0, K, 3 * N_LOC)]
    be = OracleShardBackend(cb.poses, cb.embeddings, cb.mesh_vertices)

    class FakeComm:
        def __init__(self, r):
            self.rank, self.world = r, 3

    engs = [ShardedFilterEngine(num_particles=N_LOC, backend=be, comm=FakeComm(r)) for r in range(3)]
    for r, e in enumerate(engs):
        e.set_particles(torch.as_tensor(start[r * N_LOC:(r + 1) * N_LOC]))
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    poses, N = start, 3 * N_LOC
    for t in range(1, 3):
        run_lockstep(engs, [((torch.as_tensor(traj.odoms[t]), torch.as_tensor(traj.codes[t])), {}) for _ in engs])
        tn, rot = oracle.philox_noise(N, 4000, t - 1, np.float32(2e-4), np.float32(0.5))
        ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn, rot, u=oracle.philox_uniform64(N, 4000, t - 1))
        assert np.array_equal(np.concatenate([e.ridx.numpy() for e in engs]), ref["ridx"])
        assert np.array_equal(np.concatenate([e.weights.numpy() for e in engs]), ref["weights"])
        poses = ref["poses"]
