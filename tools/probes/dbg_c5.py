import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import oracle
from test_gpu_fullsize import _brute_force_all
from midastouch_amd.engine import BatchFilterEngine
from midastouch_amd import ops
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
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
V = cb.mesh_vertices
for t in (1, 2):
    odoms = torch.as_tensor(np.stack([trajs[b % 8].odoms[t] for b in range(B)])).to(dev)
    codes = torch.as_tensor(np.stack([trajs[b % 8].codes[t] for b in range(B)])).to(dev)
    eng.step(odoms, codes, None)
prop = eng.poses_prop.cpu().numpy().reshape(B * N, 4, 4)
nn, d3 = _brute_force_all(oracle, prop, oracle.R3_SE3(cb.poses), V)
w = eng.weights.cpu().numpy().reshape(-1)
bad = np.nonzero((w != 0) != ~(d3 > 0.002))[0]
print("bad", bad, "tele", eng.telemetry.cpu().numpy()[:4])
dd = ops.nn3_dist(eng.tree3, torch.as_tensor(prop[bad]).to(dev)).cpu().numpy()
print("standalone nn3_dist", dd, "oracle", d3[bad])
for i in bad:
    tq = prop[i][:3, 3].astype(np.float64)
    e = nn[i]
    c = cb.poses[e][:3, 3].astype(np.float64)
    delta = np.linalg.norm(tq - c)
    dv = np.linalg.norm(V - tq, axis=1)
    rho = np.linalg.norm(V - c, axis=1)
    order = np.argsort(rho, kind="stable")
    vbest = int(np.argmin(dv))
    rank = int(np.nonzero(order == vbest)[0][0])
    within = np.nonzero(dv <= 0.002)[0]
    ranks_within = sorted(int(np.nonzero(order == v)[0][0]) for v in within)
    print(f"particle {i} traj {i//N} slot {i%N} entry {e} delta {delta:.6f} dmin {dv.min():.6f} rank of nearest in entry list {rank} rho_nearest {rho[vbest]:.6f} "
          f"rho[255] {rho[order[255]]:.6f} rho[256] {rho[order[256]]:.6f} lim {0.002+delta:.6f} n_within {len(within)} ranks {ranks_within[:8]}")
    # first index where rho > lim
    lim = 0.002 + delta
    print("   first stop rank", int(np.argmax(rho[order] > lim)), " nan in pose", np.isnan(prop[i]).any(), "hint", eng.hint.cpu().numpy().reshape(-1)[i])
# a second run of the same frame from the same state?  the valid flags
val = eng._valid.cpu().numpy().reshape(-1) if hasattr(eng, "_valid") else None
print("valid flags at bad", None if val is None else val[bad])
