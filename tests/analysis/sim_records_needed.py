"""CPU analysis with the oracle (not a test, not product): records a one-pivot certificate needs on the c5 workload; see DESIGN.md section 4."""
import sys, numpy as np, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from midastouch_amd.synthetic import make_codebook, make_trajectory
from scipy.spatial import cKDTree
name = sys.argv[1] if len(sys.argv) > 1 else "cotter-pin"
cb = make_codebook(name, K=50000, D=512, seed=1005)
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
for t in range(1, 16):
    tn, rot = O.philox_noise(N, 7, t, sig_t, sig_r)
    u = O.philox_uniform64(N, 7, t)
    out = f.step(poses, tr.odoms[t], tr.codes[t], tn, rot, u=u)
    if t in (5, 15):
        q = out["feat"].astype(np.float64); nn = out["nn_idx"]; dstar = np.sqrt(out["nn_d2"].astype(np.float64))
        r = np.linalg.norm(q - F[hint], axis=1)
        sel = rng.choice(N, 1500, replace=False)
        needs, needs_ideal, needs_piv = [], [], []
        for i in sel:
            h = hint[i]
            dd, ii = tree.query(F[h], k=513)
            rho = dd  # ascending, rho[0]=0
            dq = np.linalg.norm(F[ii] - q[i], axis=1)
            best = np.minimum.accumulate(dq)
            # adaptive: first s with rho[s] - r > best[s-1]
            ok = np.nonzero(rho[1:] - r[i] > best[:-1])[0]
            needs.append(ok[0] + 1 if len(ok) else 513)
            ok2 = np.nonzero(rho - r[i] > dstar[i])[0]
            needs_ideal.append(ok2[0] if len(ok2) else 513)
            needs_piv.append(len(tree.query_ball_point(F[nn[i]], 2 * dstar[i])))
        needs = np.array(needs); ni = np.array(needs_ideal); npv = np.array(needs_piv)
        pct = lambda a: [int(np.percentile(a, p)) for p in (10, 25, 50, 75, 90, 97, 99)]
        print(f"frame {t}: distinct hints {len(np.unique(hint))}, median r {np.median(r)*1e3:.3f} mm, median d* {np.median(dstar)*1e3:.3f} mm, hint==nn {np.mean(hint==nn):.2f}")
        print("  records needed (adaptive) p10..p99", pct(needs), "mean", needs.mean())
        print("  ideal single pivot", pct(ni), "mean", ni.mean())
        print("  pivot at true NN, radius 2d*", pct(npv), "mean", npv.mean())
        print("  frac <=32:", np.mean(needs <= 32), " <=64:", np.mean(needs<=64), " <=96:", np.mean(needs <= 96), "<=160", np.mean(needs<=160))
    poses, hint = out["poses"], out["nn_idx_res"]
