import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from midastouch_amd.synthetic import make_codebook, make_trajectory
from oracle.ref_shaped import RefShapedFilter
cb = make_codebook(K=50000, D=512, seed=1001); traj = make_trajectory(cb, T=8, seed=2001)
N = 100000
rng = np.random.default_rng(0)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    flt = RefShapedFilter(cb.poses, cb.embeddings, cb.mesh_vertices, workers=nt)
    poses = torch.as_tensor(cb.poses[rng.integers(0, cb.K, N)])
    od, codes = torch.as_tensor(traj.odoms), torch.as_tensor(traj.codes)
    poses, _ = flt.step(poses, od[1], codes[1][None])
    t0 = time.perf_counter()
    for t in (2, 3):
        poses, _ = flt.step(poses, od[t], codes[t][None])
    print(nt, "threads:", (time.perf_counter() - t0) / 2, "s/step", flush=True)
