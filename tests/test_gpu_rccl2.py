"""Two real ranks (one process each): the particle-sharded frame, every exchange form, against the ORACLE's frame of all
particles.  Over RCCL with one GPU per process when the box has two MI355X (skips itself otherwise: the driver's 1-GPU
tier); and on ONE GPU shared by the two processes with gloo carrying the exchanges - the whole multi-process path
(HipShardBackend kernels, TorchDistComm, the exchange protocol) except RCCL's transport.  "peer_c" is the C-side frame
(midas_shard_step): over RCCL one call per frame on the library's own communicator; on the shared GPU two calls around the
gloo gather of the records, the rows and the completion flags going through the interprocess-mapped inboxes either way."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

N_LOC, K, D, FRAMES, SEED = 8192, 4000, 256, 6, 4000


def _data():
    from midastouch_amd.synthetic import make_codebook, make_trajectory
    cb = make_codebook(K=K, D=D, seed=1000)
    traj = make_trajectory(cb, T=FRAMES + 1, seed=2000)
    start = cb.poses[np.random.default_rng(0).integers(0, K, 2 * N_LOC)]
    return cb, traj, start


def _worker(rank, world, port, exchange, out_dir, shared_gpu=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0 if shared_gpu else rank)
    torch.cuda.set_device(dev)
    if shared_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from midastouch_amd.dist import ShardedFilterEngine
    cb, traj, start = _data()
    eng = ShardedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N_LOC, seed=SEED, device=dev, exchange=exchange)
    eng.set_particles(torch.as_tensor(start[rank * N_LOC:(rank + 1) * N_LOC]))
    res = []
    for t in range(1, FRAMES + 1):
        eng.step(torch.as_tensor(traj.odoms[t]).to(dev), torch.as_tensor(traj.codes[t]).to(dev), gt=torch.as_tensor(traj.gt_poses[t]).to(dev))
        res.append({k: getattr(eng, k).cpu().numpy().copy() for k in ("nn_idx", "weights", "ridx", "poses", "status", "rmse")})
    res[0]["exchange"] = eng.exchange
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _run_and_check(tmp_path, oracle, exchange, shared_gpu):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, exchange, str(tmp_path), shared_gpu), nprocs=2, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), f"r{r}.pt"), weights_only=False) for r in range(2)]
    if exchange != "auto":
        assert all(p[0]["exchange"] == exchange for p in parts)
    else:  # under RCCL "auto" maps the inboxes when the start-up self test passes on both ranks
        assert parts[0][0]["exchange"] == parts[1][0]["exchange"]
    cb, traj, start = _data()
    ofl = oracle.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
    poses = start
    for t in range(1, FRAMES + 1):
        tn, rot = oracle.philox_noise(2 * N_LOC, SEED, t - 1, np.float32(2e-4), np.float32(0.5))
        ref = ofl.step(poses, traj.odoms[t], traj.codes[t], tn, rot, u=oracle.philox_uniform64(2 * N_LOC, SEED, t - 1))
        got = lambda k: np.concatenate([p[t - 1][k] for p in parts])  # noqa: E731
        assert np.array_equal(got("nn_idx"), ref["nn_idx"]), t
        np.testing.assert_allclose(got("weights"), ref["weights"], rtol=1e-12, atol=0)
        assert np.array_equal(got("ridx"), ref["ridx"]), t
        assert np.array_equal(got("poses"), ref["poses"]), t
        poses = ref["poses"]


@pytest.mark.parametrize("exchange", ["peer_c", "peer", "auto", "a2a_fixed", "a2a", "allgather"])
def test_two_ranks_over_rccl_match_the_oracle(tmp_path, oracle, exchange):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _run_and_check(tmp_path, oracle, exchange, shared_gpu=False)


@pytest.mark.parametrize("exchange", ["peer_c", "peer", "a2a_fixed", "a2a", "allgather"])
def test_two_processes_sharing_one_gpu_match_the_oracle(tmp_path, oracle, exchange):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _run_and_check(tmp_path, oracle, exchange, shared_gpu=True)
