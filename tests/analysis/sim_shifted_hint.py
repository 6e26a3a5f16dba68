"""CPU analysis with the oracle (not a test, not product): does pushing every hinted ENTRY through the frame's odometry (a per-frame map
hinted entry -> predicted entry, no noise) give the list scan a better pivot?  c5 workload; see DESIGN.md section 4."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from midastouch_amd.synthetic import make_codebook, make_trajectory
from scipy.spatial import cKDTree
M = 512
cb = make_codebook("cotter-pin", K=50000, D=512, seed=1005)
tr = make_trajectory(cb, T=40, seed=2200)
N = 10000
rng = np.random.default_rng(1)
d0 = np.linalg.norm(cb.poses[:, :3, 3] - tr.gt_poses[0][:3, 3], axis=1)
poses = cb.poses[rng.choice(np.argsort(d0)[:2500], N)].astype(np.float32)
f = O.OracleFilter(cb.poses, cb.embeddings, cb.mesh_vertices)
F = f.cb_feat.astype(np.float64)
tree = cKDTree(F)
hint = f.SE3_NN_idx(poses)
sig_t, sig_r = 1e-4, 0.5
def needed(h, qi):
    rho, ii = tree.query(F[h], k=M + 1)
    dq = np.linalg.norm(F[ii] - qi, axis=1)
    r = np.linalg.norm(qi - F[h])
    best = np.minimum.accumulate(dq)[:-1]
    ok = np.nonzero(rho[1:] - r > best)[0]
    return ok[0] + 1 if len(ok) else M + 1
def scanned(n):  # granularity of the kernel: 32 solo, then chunks of 64
    return 32 if n <= 32 else 32 + 64 * int(np.ceil((n - 32) / 64))
zero3 = np.zeros((1, 3), np.float32)
for t in range(1, 26):
    tn, rot = O.philox_noise(N, 7, t, sig_t, sig_r)
    u = O.philox_uniform64(N, 7, t)
    out = f.step(poses, tr.odoms[t], tr.codes[t], tn, rot, u=u)
    if t in (5, 10, 15, 20, 25):
        q = out["feat"].astype(np.float64); nn = out["nn_idx"]
        # map: every distinct hinted entry pushed through the frame's odometry without noise
        uh = np.unique(hint)
        moved = O.propagate(cb.poses[uh].astype(np.float32), tr.odoms[t], np.zeros((len(uh), 3), np.float32), np.zeros((len(uh), 3), np.float32))
        fm = O.R3_SE3(moved).astype(np.float64)
        mp = dict(zip(uh.tolist(), tree.query(fm)[1].tolist()))
        piv = np.array([mp[h] for h in hint.tolist()])
        sel = rng.choice(N, 800, replace=False)
        a = np.array([needed(hint[i], q[i]) for i in sel]); b = np.array([needed(piv[i], q[i]) for i in sel]); c = np.array([needed(nn[i], q[i]) for i in sel])
        sa, sb, sc = [np.mean([scanned(x) for x in v]) for v in (a, b, c)]
        print(f"frame {t}: distinct hints {len(uh)}; hint==nn {np.mean(hint==nn):.2f} map==nn {np.mean(piv==nn):.2f} | needed hint {a.mean():.1f} map {b.mean():.1f} ideal {c.mean():.1f} | scanned hint {sa:.0f} map {sb:.0f} ideal {sc:.0f} | r hint {np.median(np.linalg.norm(q-F[hint],axis=1))*1e3:.2f} map {np.median(np.linalg.norm(q-F[piv],axis=1))*1e3:.2f} mm")
    poses, hint = out["poses"], out["nn_idx_res"]
