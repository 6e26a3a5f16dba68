"""CPU analysis with the oracle (not a test, not product): records needed by a one-pivot certificate that splits the query offset
into a part inside a per-entry 'normal' subspace and the rest (Cauchy-Schwarz per part) against the plain triangle inequality."""
import sys, numpy as np, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from midastouch_amd.synthetic import make_codebook, make_trajectory
from scipy.spatial import cKDTree
name = sys.argv[1] if len(sys.argv) > 1 else "cotter-pin"
M = 512
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
def needed(rho, dq, r, reff_s):
    """first s >= 1 such that records s.. cannot beat best of records < s"""
    best = np.minimum.accumulate(dq)[:-1]          # best over records 0..s-1, for s = 1..
    rs = rho[1:]; re = reff_s[1:]
    fv = rs * rs - 2 * rs * re + r * r - best * best
    ok = np.nonzero((rs >= re) & (fv > 0))[0]
    return ok[0] + 1 if len(ok) else len(rho)
for t in range(1, 16):
    tn, rot = O.philox_noise(N, 7, t, sig_t, sig_r)
    u = O.philox_uniform64(N, 7, t)
    out = f.step(poses, tr.odoms[t], tr.codes[t], tn, rot, u=u)
    if t in (5, 15):
        q = out["feat"].astype(np.float64); nn = out["nn_idx"]
        r = np.linalg.norm(q - F[hint], axis=1)
        sel = rng.choice(N, 1500, replace=False)
        res = {}
        for i in sel:
            h = hint[i]
            rho, ii = tree.query(F[h], k=M + 1)
            V = F[ii] - F[h]
            dq = np.linalg.norm(F[ii] - q[i], axis=1)
            dl = q[i] - F[h]
            res.setdefault("plain", []).append(needed(rho, dq, r[i], np.full(M + 1, r[i])))
            for mp in (32, 128):
                w, E = np.linalg.eigh(V[1:mp + 1].T @ V[1:mp + 1])   # ascending eigenvalues
                for k in (1, 2, 3):
                    Nk = E[:, :k]
                    a = np.linalg.norm(V @ Nk, axis=1)
                    ratio = np.zeros(M + 1); ratio[1:] = a[1:] / rho[1:]
                    alpha = np.maximum.accumulate(ratio[::-1])[::-1]           # suffix max
                    dn = np.linalg.norm(dl @ Nk); dt = np.sqrt(max(r[i] ** 2 - dn ** 2, 0.0))
                    # tighter: |v_T| <= rho sqrt(1 - amin^2) ignored; use rho
                    res.setdefault(f"pca{mp}_k{k}", []).append(needed(rho, dq, r[i], dn * alpha + dt))
        pct = lambda a: [int(np.percentile(a, p)) for p in (10, 25, 50, 75, 90, 97, 99)]
        print(f"frame {t}: median r {np.median(r)*1e3:.3f} mm hint==nn {np.mean(hint==nn):.2f}")
        for k, v in res.items():
            v = np.array(v); print(f"  {k:12s} p10..p99 {pct(v)} mean {v.mean():.1f}  <=8 {np.mean(v<=8):.2f} <=16 {np.mean(v<=16):.2f} <=32 {np.mean(v<=32):.2f}")
    poses, hint = out["poses"], out["nn_idx_res"]
