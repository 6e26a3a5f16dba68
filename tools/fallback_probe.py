#!/usr/bin/env python3
"""Which particles still need the tree search (GPU box only)?  Runs the bench trajectory and dumps, for the lanes
the hint scan could not certify, the distance to the hinted entry, the NN distance and the list radius."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import ops
from midastouch_amd.engine import FilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
N, K, D = 100_000, 50_000, 512
cb = make_codebook(K=K, D=D, seed=1001)
traj = make_trajectory(cb, T=140, seed=2001)
eng = FilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
rng = np.random.default_rng(100)
d0 = np.linalg.norm(cb.poses[:, :3, 3] - traj.gt_poses[0][:3, 3], axis=1)
near = np.argsort(d0)[: max(64, K // 20)]
eng.set_particles(torch.as_tensor(cb.poses[rng.choice(near, N)]))
eng.project_to_codebook()
odoms, codes = torch.as_tensor(traj.odoms).to(dev), torch.as_tensor(traj.codes).to(dev)
feat_cb = eng.cb_feat
for t in range(1, 130):
    eng.step(odoms[t], codes[t])
    if t % 16 == 0:
        feat = ops.se3_feature(eng.poses_prop)
        hint = eng.hint_next
        lv, nd = ops.nn6_stats(eng.tree6, feat, hint)
        fb = (nd >= 0) & ((lv > 0) | (nd > 0))
        idx, d2 = ops.nn6(eng.tree6, feat, None, want_d2=True)
        r = (feat - feat_cb[hint.long().clamp(min=0)]).norm(dim=1)
        wnorm = feat[:, 3:].norm(dim=1) / 0.01
        sel = fb.nonzero().flatten()[:8]
        print(f"frame {t}: fallback lanes {int(fb.sum())}  kept {int(eng.status[1])}")
        for i in sel.tolist():
            print(f"   lane {i}: r_hint={float(r[i])*1e3:.2f} mm  d_nn={float(d2[i].sqrt())*1e3:.2f} mm  |w|={float(wnorm[i]):.3f} rad"
                  f"  hint={int(hint[i])} nn={int(idx[i])}  w_hint={float(feat_cb[hint[i].long(),3:].norm()/0.01):.3f}")
