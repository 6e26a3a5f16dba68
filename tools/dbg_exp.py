import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from midastouch_amd import engine as E
from midastouch_amd.synthetic import make_codebook, make_trajectory
from oracle import oracle as orc
dev = torch.device('cuda', 0)
for N in (100_000, 1_000_000):
  for name in ("FilterEngine", "PipelinedFilterEngine"):
    K, D, seed = 50_000, 512, 4000
    cb = make_codebook("035_power_drill", K=K, D=D, seed=1003)
    traj = make_trajectory(cb, T=6, seed=2003)
    eng = getattr(E, name)(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=seed, device=dev)
    rng = np.random.default_rng(0)
    eng.set_particles(torch.as_tensor(cb.poses[rng.integers(0, K, N)]))
    eng.step(torch.as_tensor(traj.odoms[1]), torch.as_tensor(traj.codes[1]), gt=torch.as_tensor(traj.gt_poses[1]))
    nn = eng.nn_idx.cpu().numpy()
    w = eng.weights.cpu().numpy()
    mask = w != 0
    scores = orc.score_codebook(cb.embeddings, traj.codes[1])
    sc_dev = eng.codebook.score(torch.as_tensor(traj.codes[1]).to(dev))[0].cpu().numpy()
    print(N, name, "scores equal:", np.array_equal(scores, sc_dev), "rows differing", int((scores != sc_dev).sum()))
    e = orc.exp_spec(scores[nn], 1.0)
    S = orc.blocked_scan(e)[1]
    ref = e / S * mask
    bad = np.flatnonzero(w != ref)
    print("   weights mismatches", len(bad), "of", N, "max rel", float(np.max(np.abs(w - ref)[mask] / ref[mask])))
    if len(bad):
        i = bad[0]
        print("   first bad", i, w[i].hex(), ref[i].hex(), "S implied", (e[i] / w[i]), "S", S)
        # is it e or S?  ratio w/ref over the kept particles
        r = w[mask] / ref[mask]
        print("   ratio min/max", r.min(), r.max(), "distinct ratios", len(np.unique(r)))
