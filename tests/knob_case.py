"""One fixed scenario through every engine form whose kernels the library's MIDAS_* environment switches choose between; prints one
sha256 line per part.  tests/test_gpu_knobs.py runs it in a fresh process per switch (the library reads most of them once) and
compares the lines with the default's: every alternative path has to give the same bits.

Parts: `pipelined` (PipelinedFilterEngine, N = 100 000: the single-launch front with per-wave tables, the grouped tail, the
prediction list, `run()` and `step()`), `eager` (FilterEngine: the unfolded tail and resample), `batch` (PipelinedBatchFilterEngine:
the presorted front), `loop` (the reference-named loop with DBSCAN, cluster centres and annealing, from 60 000 particles down to
the small-set kernels).  Needs an MI355X."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def digest(*tensors) -> str:
    h = hashlib.sha256()
    for t in tensors:
        a = t.detach().cpu().contiguous().numpy() if isinstance(t, torch.Tensor) else np.ascontiguousarray(t)
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def main():
    from midastouch_amd.engine import FilterEngine, PipelinedBatchFilterEngine, PipelinedFilterEngine
    from midastouch_amd.synthetic import make_codebook, make_trajectory

    dev = torch.device("cuda", 0)
    K, D = 12000, 256
    cb = make_codebook("004_sugar_box", K=K, D=D, seed=1301)
    traj = make_trajectory(cb, T=40, seed=2301)
    od, co, gt = (torch.as_tensor(x).to(dev) for x in (traj.odoms, traj.codes, traj.gt_poses))
    rng = np.random.default_rng(5)

    # pipelined, full-size particle set: wide start (many rows claimed), run() then step()
    N = 100_000
    eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4100, device=dev)
    eng.set_particles(torch.as_tensor(cb.poses[rng.integers(0, K, N)]))
    eng.project_to_codebook()
    eng.run(od[1:13], co[1:13], gts=gt[1:13])
    for t in range(13, 18):
        eng.step(od[t], co[t], gt=gt[t])
    print("pipelined", digest(eng.nn_idx, eng.poses_prop, eng.ridx, eng.poses, eng.weights, eng.weights_res, eng.hint, eng.status, eng.rmse))

    # eager engine, ragged size
    N2 = 30_011
    eg = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N2, seed=4200, device=dev)
    eg.set_particles(torch.as_tensor(cb.poses[rng.integers(0, K, N2)]))
    for t in range(1, 8):
        eg.step(od[t], co[t], gt=gt[t])
    print("eager", digest(eg.nn_idx, eg.poses_prop, eg.ridx, eg.poses, eg.weights, eg.status, eg.rmse))

    # batch of trajectories (presorted front; the batch engine takes sparse scoring only)
    if os.environ.get("MIDAS_DENSE_SCORES"):
        return loop_part(cb, traj, dev, K, D, gt, co)
    B, Nb = 6, 5000
    be = PipelinedBatchFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, B, Nb, seed=4300, device=dev)
    be.set_particles(torch.as_tensor(np.stack([cb.poses[rng.integers(0, K, Nb)] for _ in range(B)])))
    be.project_to_codebook()
    ob = torch.stack([od] * B, dim=1)
    cbb = torch.stack([co] * B, dim=1)
    for t in range(1, 9):
        be.step(ob[t], cbb[t])
    print("batch", digest(be.nn_idx, be.poses_prop, be.ridx, be.poses, be.weights, be.status))
    loop_part(cb, traj, dev, K, D, gt, co)


def loop_part(cb, traj, dev, K, D, gt, co):
    # the reference-named loop: DBSCAN in frame 0, annealing every frame, the set shrinks into the small-set kernels
    from midastouch_amd.config import load_config
    from midastouch_amd.filter import Sequence, filter as run_filter
    from midastouch_amd.tactile_tree import tactile_tree

    torch.manual_seed(77)  # (init_filter draws on the host)
    np.random.seed(77)
    cfg = load_config(["expt.params.num_particles=60000", f"expt.codebook_size={K}", f"tcn.model.output_dim={D}"])
    tree = tactile_tree(torch.as_tensor(cb.poses), torch.as_tensor(cb.cam_poses), torch.as_tensor(cb.embeddings))
    tree.to_device(dev)
    seq = Sequence(gt, torch.as_tensor(traj.meas_poses).to(dev), co, tree, cb.mesh_vertices, "004_sugar_box")
    st = run_filter(cfg, seq, device=dev, floor=1000, max_frames=36, seed=4400)
    flat = lambda xs: np.concatenate([np.asarray(x, dtype=np.float64).reshape(-1) for x in xs])  # noqa: E731
    print("loop", digest(np.asarray(st["num_particles"], dtype=np.int64), np.asarray(st["rmse_t"]), np.asarray(st["rmse_r"]),
                         flat(st["cluster_poses"]), flat(st["cluster_stds"])), "N", st["num_particles"][-1])


if __name__ == "__main__":
    main()
