import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from midastouch_amd.dist import ShardedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
cb = make_codebook("004_sugar_box", K=50000, D=512, seed=1001); tr = make_trajectory(cb, T=300, seed=2001)
for ex in ("allgather", "a2a_fixed", "peer", "peer_c"):
    eng = ShardedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, 100000, seed=4000, device=dev, exchange=ex)
    d0 = np.linalg.norm(cb.poses[:, :3, 3] - tr.gt_poses[0][:3, 3], axis=1)
    eng.set_particles(torch.as_tensor(cb.poses[np.random.default_rng(0).choice(np.argsort(d0)[:2500], 100000)])); eng.project_to_codebook()
    od, co = torch.as_tensor(tr.odoms).to(dev), torch.as_tensor(tr.codes).to(dev)
    for i in range(20): eng.step(od[1 + i], co[1 + i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200): eng.step(od[21 + i], co[21 + i])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(ex, eng.exchange, "host enqueue us/step %.1f, total us/step %.1f" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
    if eng.exchange == "peer_c" and eng._ccomm is not None:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.run(od[21:221], co[21:221])
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(ex, "midas_shard_run(200): host enqueue us/step %.1f, total us/step %.1f" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(100): eng.step(od[21 + i], co[21 + i])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
