#!/usr/bin/env python3
"""Wall-clock stamps (100 MHz) of one frame of the pipelined engine on the bench workload, steady state (debug): where the
front's particle / scoring waves and the grouped tail's waves start and end, and the phases of two tail waves.
Needs a library built with -DMIDAS_DEBUG_CLOCKS:  tools/variants.sh dbg "-DMIDAS_DEBUG_CLOCKS" ;
MIDAS_HIP_LIB=$PWD/midastouch_amd/csrc/build/variants/dbg.so python tools/tg_clocks.py"""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import _lib
from midastouch_amd.engine import PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
N, K, D = 100_000, 50_000, 512
cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
T = 140
traj = make_trajectory(cb, T=T, seed=2001)
eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4000, device=dev)
rng = np.random.default_rng(100)
d0 = np.linalg.norm(cb.poses[:, :3, 3] - traj.gt_poses[0][:3, 3], axis=1)
near = np.argsort(d0)[: max(64, K // 20)]
eng.set_particles(torch.as_tensor(cb.poses[rng.choice(near, N)]))
eng.project_to_codebook()
odoms, codes = (torch.as_tensor(x).to(dev) for x in (traj.odoms, traj.codes))
lib = _lib.load()
tg, tw, ff = (ctypes.c_longlong * 64)(), (ctypes.c_longlong * 8192)(), (ctypes.c_longlong * 16384)()
rows, pairs = [], []
NPU = -(-N // 64)  # one-wave particle workgroups of the front (the scoring / list workgroups follow)
gts = torch.as_tensor(traj.gt_poses).to(dev)
for t in range(1, 41):
    eng.step(odoms[t], codes[t])
torch.cuda.synchronize()
def front_span(f):
    f = f.reshape(-1, 2)
    fp, fs = f[:NPU], f[NPU:][f[NPU:, 0] > 0]
    z = fp[:, 0].min()
    return [fp[:, 0].min(), fp[:, 0].max(), fp[:, 1].min(), np.median(fp[:, 1]), fp[:, 1].max(), fs[:, 0].min() if len(fs) else z, fs[:, 1].max() if len(fs) else z]
def tail_span(w):
    g = w[:2 * -(-N // 256)].reshape(-1, 2)
    pl = w[2050:].reshape(-1, 2); pl = pl[pl[:, 0] > 0]
    z = g[:, 0].min()
    return [g[:, 0].min(), g[:, 0].max(), g[:, 1].min(), np.median(g[:, 1]), g[:, 1].max(), w[2048], w[2049], pl[:, 0].min() if len(pl) else z, pl[:, 1].max() if len(pl) else z]
for t in range(41, T - 4, 2):
    lib.midas_debug_tg_clocks(tg, 1); lib.midas_debug_ff_clocks(ff, 1)
    eng.run(odoms[t:t + 2], codes[t:t + 2], gts[t:t + 2])  # two frames by one call: the hand-over between them is the one the bench times
    torch.cuda.synchronize()
    lib.midas_debug_tg_clocks(tg, 0); lib.midas_debug_tg_waves(tw); lib.midas_debug_ff_clocks(ff, 0)
    w, f = np.array(tw[:], dtype=np.float64), np.array(ff[:], dtype=np.float64)
    fr = sorted([front_span(f[:8192]), front_span(f[8192:])])  # (the two frames' stamps are kept apart by parity: in time order)
    tl = sorted([tail_span(w[:4096]), tail_span(w[4096:])])
    t0 = fr[0][0]
    rows.append((np.array(fr[0] + tl[0] + fr[1] + tl[1]) - t0) / 100.0)
r = np.median(np.array(rows), axis=0)
print("us from the start of the first frame's first particle wave (medians over %d pairs of frames by one midas_lazy_run call)" % len(rows))
for nm, v in (("frame A", r[:16]), ("frame B", r[16:])):
    print(nm, "front particle waves: start %.2f .. %.2f ; end first %.2f median %.2f last %.2f ; scoring / list waves %.2f .. %.2f" % tuple(v[:7]))
    print(nm, "tail group waves: start %.2f .. %.2f ; end first %.2f median %.2f last %.2f ; rmse workgroup %.2f .. %.2f ; list workgroups %.2f .. %.2f" % tuple(v[7:16]))
print("frame period %.2f us ; front end -> tail start %.2f ; tail end (incl. rmse / list workgroups) -> next front start %.2f" % (r[16] - r[0], r[7] - max(r[4], r[6]), r[16] - max(r[11], r[13], r[15])))
