#!/usr/bin/env python3
"""The frames the driver's flags time (frames 8..27 after the wide start of bench.py) against the steady state, frame by
frame: HIP-event time of the frame, rows off the prediction list / claimed by particle waves, tree fall-backs, and - with a
library built with -DMIDAS_DEBUG_CLOCKS (tools/variants.sh dbg "-DMIDAS_DEBUG_CLOCKS") - when the front's particle waves
and its list waves end.   MIDAS_HIP_LIB=$PWD/midastouch_amd/csrc/build/variants/dbg.so python tools/diag_early.py"""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import _lib
from midastouch_amd.engine import PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory, wide_start
dev = torch.device("cuda", 0)
N, K, D = 100_000, 50_000, 512
cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
T = 140
traj = make_trajectory(cb, T=T, seed=2001)
eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4000, device=dev)
eng.set_particles(torch.as_tensor(wide_start(cb.extents, traj.gt_poses[0], N, 100)))
eng.project_to_codebook()
odoms, codes, gts = (torch.as_tensor(x).to(dev) for x in (traj.odoms, traj.codes, traj.gt_poses))
lib = _lib.load()
dbg = hasattr(lib, "midas_debug_ff_clocks")
ff = (ctypes.c_longlong * 16384)()
NPU = -(-N // 64)
print("frame   us   list  claimed  nn_fb prune_fb  distinct_nn  valid | particle waves end: first median p90 last ; list waves end (us from first wave start)")
prev = eng.telemetry.cpu().numpy().copy()
for t in range(1, 100):
    if dbg:
        lib.midas_debug_ff_clocks(ff, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.step(odoms[t], codes[t], gt=gts[t])
    e1.record()
    torch.cuda.synchronize()
    tel = eng.telemetry.cpu().numpy().copy()
    d = tel - prev
    prev = tel
    eng.flush()
    nn = eng._nn[eng._cur].cpu().numpy()
    extra = ""
    if dbg:
        lib.midas_debug_ff_clocks(ff, 0)
        f = np.array(ff[:], dtype=np.float64)
        for half in (f[:8192], f[8192:]):
            h = half.reshape(-1, 2)
            fp = h[:NPU]
            if fp[:, 0].max() <= 0:
                continue
            fs = h[NPU:][h[NPU:, 0] > 0]
            z = fp[:, 0].min()
            en = (fp[:, 1] - z) / 100.0
            extra = " | %.1f %.1f %.1f %.1f ; %.1f (start spread %.1f)" % (en.min(), np.median(en), np.percentile(en, 90), en.max(), (fs[:, 1].max() - z) / 100.0 if len(fs) else 0.0, (fp[:, 0].max() - z) / 100.0)
    if t < 40 or t % 10 == 0:
        print("%4d %6.1f %6d %6d %6d %6d %8s %8s%s" % (t, 1e3 * e0.elapsed_time(e1), d[3], d[2], d[0], d[1],
              len(np.unique(nn[:N])), int(eng._valid.sum().item()), extra))
