#!/usr/bin/env python3
"""The diffuse regime frame by frame: the first frames after init_filter(gt_0, N) + projection (bench.py's start, what the
driver's --warmup 5 --steps 20 window times).  Per frame: HIP-event time, distinct nearest entries, particles kept, and - with
MIDAS_ABLATE=4 - the per-wave phase clocks and cooperative-lane counts of the front kernel.
usage: [MIDAS_ABLATE=4] [MIDAS_DENSE_SCORES=1] tools/diag_diffuse.py [frames]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.engine import PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory, mesh_scale
from scipy.spatial.transform import Rotation

dev = torch.device("cuda", 0)
N, K, D = 100000, 50000, 512
T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
stats = int(os.environ.get("MIDAS_ABLATE", "0")) & 4
cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
traj = make_trajectory(cb, T=T + 4, seed=2001)
eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, N, seed=4000, device=dev)
od, co, gt = (torch.as_tensor(x).to(dev) for x in (traj.odoms, traj.codes, traj.gt_poses))


def wide_init(seed):
    g = torch.Generator().manual_seed(seed)
    tn0 = torch.normal(0.0, mesh_scale(cb.extents) / 3.0, size=(N, 3), generator=g)
    rn0 = torch.normal(0.0, 60.0, size=(N, 3), generator=g)
    Tn = torch.zeros((N, 4, 4))
    Tn[:, :3, :3] = torch.as_tensor(Rotation.from_euler("zyx", rn0.numpy(), degrees=True).as_matrix()).float()
    Tn[:, :3, 3], Tn[:, 3, 3] = tn0, 1.0
    eng.set_particles(torch.as_tensor(traj.gt_poses[0])[None] @ Tn)
    eng.project_to_codebook()


wide_init(100)
for t in range(1, 3):
    eng.step(od[t], co[t], gt=gt[t])
torch.cuda.synchronize()
wide_init(200)
nw = (N + 63) // 64
rows = []
prev = eng.telemetry[16:].view(-1, 16).clone() if stats else None
for t in range(1, T + 1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.step(od[t], co[t], gt=gt[t])
    e1.record()
    torch.cuda.synchronize()
    uniq = int(torch.unique(eng.nn_idx).numel())
    kept = int(eng._valid.sum().item())
    line = {"frame": t, "us": round(1e3 * e0.elapsed_time(e1), 1), "distinct_nn": uniq, "kept": kept}
    if stats:
        cur = eng.telemetry[16:].view(-1, 16).clone()
        d = (cur - prev)[:nw].cpu().numpy().astype(float)
        prev = cur
        life = d[:, 7] / 100.0
        start = cur[:nw, 1].cpu().numpy().astype(float)
        start = (start - start.min()) / 100.0  # us after the first wave's start
        if t in (2, 3, 6, 10, 20, 30):
            order = np.argsort(life)
            names = ["resample+prop", "nn_solo", "nn_coop+tree", "claim+score", "prune_lists", "tree3", "gather", "reduce"]
            print("   rows scored per wave: mean %.1f max %d; corr(life, rows) %.2f corr(life, nn_coop) %.2f corr(life, start) %.2f" % (
                d[:, 0].mean(), d[:, 0].max(), np.corrcoef(life, d[:, 0])[0, 1], np.corrcoef(life, d[:, 10])[0, 1], np.corrcoef(life, start)[0, 1]))
            for tag, sel in (("slowest", order[-5:]), ("median", order[nw // 2 - 2: nw // 2 + 2])):
                for w in sel:
                    print(f"   {tag} wave {w}: start {start[w]:.1f} life {life[w]:.1f} us rows {int(d[w,0])} nn_coop_lanes {int(d[w,2])} mesh_coop_lanes {int(d[w,3])} | kticks " +
                          " ".join(f"{n}={d[w, 8 + i] / 1e3:.1f}" for i, n in enumerate(names)))
            end = start + life
            print("   kernel span %.1f us; waves ending in the last 10 us: %d; start p50 %.1f max %.1f" % (end.max(), int((end > end.max() - 10).sum()), np.median(start), start.max()))
        line.update(rows_scored=int(d[:, 0].sum()), nn_coop_lanes=int(d[:, 2].sum()), mesh_coop_lanes=int(d[:, 3].sum()), scanned_per_particle=round(d[:, 6].sum() / N, 1),
                    wave_us_p50=round(float(np.median(life)), 1), wave_us_max=round(float(life.max()), 1),
                    kticks_mean=dict(zip(["resample+prop", "nn_solo", "nn_coop+tree", "claim+score", "prune_lists", "tree3", "gather", "reduce"],
                                         (d[:, 8:16].mean(0) / 1e3).round(1).tolist())))
    rows.append(line)
    print(line, flush=True)
us = np.array([r["us"] for r in rows])
print("mean us frames 1..20: %.1f   frames 6..25: %.1f   first: %.1f" % (us[:20].mean(), us[5:25].mean() if T >= 25 else float("nan"), us[0]))
print("fallbacks (nn tree, prune tree):", eng.telemetry[:2].tolist())
