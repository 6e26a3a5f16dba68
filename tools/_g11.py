import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from midastouch_amd.engine import PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
N, K = 5000, 3000
cb = make_codebook("004_sugar_box", K=K, D=128, seed=1022); traj = make_trajectory(cb, T=20, seed=2022)
start = torch.as_tensor(cb.poses[np.random.default_rng(5).integers(0, K, N)])
od, co, gt = (torch.as_tensor(x).to(dev) for x in (traj.odoms, traj.codes, traj.gt_poses))
out = []
for flag in ("1", "0", "1"):
    os.environ["MIDAS_GUIDE"] = flag
    eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=91, softmax=False, device=dev)
    eng.set_particles(start)
    rec = []
    for t in range(1, 9):
        eng.step(od[t], co[t], gt=gt[t])
        if t > 1:
            rec.append(eng._ridx.cpu().numpy().copy())
        if t == 3 and flag == "1":
            Np = -(-N // 16) * 16; tb = eng._tables.cpu().numpy()
            lp = tb[3 * Np:4 * Np][:N]
            print("lp_raw monotone:", bool(np.all(np.diff(lp[:4096]) >= 0)), "min diff", np.diff(lp[:4096]).min(), "scores min", float(eng._scores.min()))
    out.append(rec)
for i, (a, b, c) in enumerate(zip(*out)):
    print("frame", i + 2, "guide vs off mismatches:", int((a != b).sum()), " guide vs guide:", int((a != c).sum()), (np.nonzero(a != b)[0][:5], a[a != b][:5], b[a != b][:5]))
