#!/usr/bin/env python3
"""Build time of the per-entry lists (neighbour graph + mesh-vertex lists): device builder vs the host builder
(MIDAS_HOST_INDEX=1), at c2's and c4's codebook sizes.  usage: tools/bench_index_build.py [K ...]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import ops
from midastouch_amd.synthetic import make_codebook, r3_se3_host
dev = torch.device("cuda", 0)
for K in [int(a) for a in sys.argv[1:]] or [50000, 500000]:
    cb = make_codebook("025_mug" if K > 100000 else "004_sugar_box", K=K, D=64, seed=1004, mode="iid")
    feat = torch.as_tensor(r3_se3_host(cb.poses).astype(np.float32)).to(dev)
    verts, poses = torch.as_tensor(cb.mesh_vertices).to(dev), torch.as_tensor(cb.poses).to(dev)
    res = {"K": K, "mesh_vertices": int(verts.shape[0])}
    keep = {}
    for tag, host in (("device", "0"), ("host", "1")):
        os.environ["MIDAS_HOST_INDEX"] = host
        torch.cuda.synchronize(); t0 = time.time()
        t6 = ops.Tree(feat); t1 = time.time()
        t3 = ops.Tree(verts); t2 = time.time()
        t6.attach_mesh(t3, poses); torch.cuda.synchronize(); t3_ = time.time()
        res[tag] = {"tree6_s": round(t1 - t0, 3), "tree3_s": round(t2 - t1, 3), "vertex_lists_s": round(t3_ - t2, 3), "total_s": round(t3_ - t0, 3)}
        keep[tag] = (t6.export("rho_out"), t6.export("twin"), t6.export("nbrs")[:: max(1, K // 2000)].copy(), t6.export("vlist")[:: max(1, K // 2000)].copy())
        del t6, t3
    res["identical"] = all(np.array_equal(a, b) for a, b in zip(keep["device"], keep["host"]))
    print(json.dumps(res))
