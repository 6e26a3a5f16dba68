"""CPU analysis with the oracle (not a test, not product): records fetched by the solo + cooperative scan with 0 - 4 pivot hops on c5; see DESIGN.md section 4."""
import sys, numpy as np, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
from midastouch_amd.synthetic import make_codebook, make_trajectory
from scipy.spatial import cKDTree
name = sys.argv[1] if len(sys.argv) > 1 else "cotter-pin"
FR = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [15, 30]
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
lists = {}
def lst(h):
    if h not in lists:
        dd, ii = tree.query(F[h], k=513); lists[h] = (dd, ii)
    return lists[h]
def scan(q, h, maxhops, solo=32, batch=8, hop_after=8):
    """returns (records fetched solo incl. hop batches, coop chunks, hops)"""
    fetched = 0; hops = 0
    while True:
        rho, ii = lst(h)
        dq = np.linalg.norm(F[ii] - q, axis=1)
        r = dq[0]
        best = np.minimum.accumulate(dq)
        if hops < maxhops:
            b = np.argmin(dq[:hop_after])
            if b != 0:
                # certified inside the hop batch?
                ok = np.nonzero(rho[1:hop_after] - r > best[:hop_after-1])[0]
                if len(ok): return fetched + hop_after, 0, hops
                fetched += hop_after; hops += 1; h = ii[b]; continue
        ok = np.nonzero(rho[1:] - r > best[:-1])[0]
        need = ok[0] + 1 if len(ok) else 513
        if need <= solo:
            return fetched + int(np.ceil(need / batch) * batch), 0, hops
        return fetched + solo, int(np.ceil((need - solo) / 64)), hops
for t in range(1, max(FR) + 1):
    tn, rot = O.philox_noise(N, 7, t, sig_t, sig_r)
    u = O.philox_uniform64(N, 7, t)
    out = f.step(poses, tr.odoms[(t - 1) % 38 + 1], tr.codes[(t - 1) % 38 + 1], tn, rot, u=u)
    if t in FR:
        q = out["feat"].astype(np.float64); nn = out["nn_idx"]
        sel = rng.choice(N, 1000, replace=False)
        print(f"frame {t}: distinct hints {len(np.unique(hint))} hint==nn {np.mean(hint==nn):.2f}")
        for mh in (0, 1, 2, 4):
            res = np.array([scan(q[i], hint[i], mh) for i in sel])
            print(f"  maxhops {mh}: solo records {res[:,0].mean():.1f}, open after solo {np.mean(res[:,1]>0):.2f}, coop chunks {res[:,1].mean():.2f}, hops {res[:,2].mean():.2f}, total records {(res[:,0]+64*res[:,1]).mean():.1f}")
    poses, hint = out["poses"], out["nn_idx_res"]
