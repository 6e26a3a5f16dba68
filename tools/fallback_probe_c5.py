#!/usr/bin/env python3
"""Which particles of the c5 batch need the tree search?  Reconstructs trajectory 0's NN queries of a frame
(same Philox draws) and prints, for the lanes the list scans could not certify, the geometry."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import ops
from midastouch_amd.engine import BatchFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
cb = make_codebook("cotter-pin", K=50000, D=512, seed=1005)
B, N = 64, 10000
trs = [make_trajectory(cb, T=40, seed=2200 + b) for b in range(8)]
od = torch.as_tensor(np.stack([trs[b % 8].odoms for b in range(B)], axis=1)).to(dev)
co = torch.as_tensor(np.stack([trs[b % 8].codes for b in range(B)], axis=1)).to(dev)
eng = BatchFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, B, N, device=dev)
rng = np.random.default_rng(1)
eng.set_particles(torch.as_tensor(np.stack([cb.poses[rng.integers(0, 50000, N)] for _ in range(B)]))); eng.project_to_codebook()
feat_cb = eng.cb_feat
for t in range(1, 30):
    if t in (5, 20):
        poses0, hint0 = eng.poses[0].clone(), eng.hint[0].clone()
        p1 = ops.propagate(poses0, od[t][0], None, None, eng.sig_t, eng.sig_r, eng.seed, eng.step_count)
        feat = ops.se3_feature(p1)
        lv, nd = ops.nn6_stats(eng.tree6, feat, hint0)
        fb = (nd >= 0) & ((lv > 0) | (nd > 0))
        idx, d2 = ops.nn6(eng.tree6, feat, None, want_d2=True)
        r = (feat - feat_cb[hint0.long().clamp(min=0)]).norm(dim=1)
        wnorm = feat[:, 3:].norm(dim=1) / 0.01
        print(f"frame {t}: fallback lanes {int(fb.sum())} of {N}; hint<0: {int((hint0 < 0).sum())}; median r_hint {float(r.median())*1e3:.3f} mm, "
              f"median d_nn {float(d2.sqrt().median())*1e3:.3f} mm; |w| near pi (>{3.0}): {int((wnorm > 3.0).sum())}")
        rho_out = None
        for i in fb.nonzero().flatten()[:10].tolist():
            print(f"   lane {i}: r_hint={float(r[i])*1e3:.3f} mm  d_nn={float(d2[i].sqrt())*1e3:.3f} mm  |w|={float(wnorm[i]):.4f} rad"
                  f"  hint={int(hint0[i])} nn={int(idx[i])}  |w_hint|={float(feat_cb[hint0[i].long(),3:].norm()/0.01):.4f}")
    eng.step(od[t], co[t])
