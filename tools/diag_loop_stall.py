#!/usr/bin/env python3
"""Where do the rare slow frames of the reference-named loop come from (bench line: config.reference_loop_frames_per_sec, one 30 ms
frame in one of seven runs)?  Runs the loop R times as bench.py does and lists, per run, the slowest frames by the device's events next
to the host time of the same iterations: a host-side pause shows in both, a device-side one only in the events."""
import gc, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.config import load_config
from midastouch_amd.filter import Sequence, filter as run_filter
from midastouch_amd.synthetic import make_codebook, make_trajectory
from midastouch_amd.tactile_tree import tactile_tree
dev = torch.device("cuda", 0)
N, K, D, T = 100_000, 50_000, 512, 200
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
floor = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
traj = make_trajectory(cb, T=T + 2, seed=2001)
tree = tactile_tree(torch.as_tensor(cb.poses), torch.as_tensor(cb.cam_poses), torch.as_tensor(cb.embeddings))
tree.to_device(dev)
cfg = load_config([f"expt.params.num_particles={N}", f"expt.codebook_size={K}", f"tcn.model.output_dim={D}"])
seq = Sequence(torch.as_tensor(traj.gt_poses[:T]).to(dev), torch.as_tensor(traj.meas_poses[:T]).to(dev), torch.as_tensor(traj.codes[:T]).to(dev), tree,
               cb.mesh_vertices, "004_sugar_box")
def throttle():
    for f in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat"):
        try:
            d = dict(l.split() for l in open(f).read().splitlines())
            return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", d.get("throttled_time", 0)))
        except Exception:
            pass
    return (-1, -1)
gc.collect(); gc.disable()
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?", flush=True)
for r in range(R):
    th0 = throttle()
    st = run_filter(cfg, seq, device=dev, floor=floor)
    th1 = throttle()
    print(f"run {r}: cgroup throttled periods +{th1[0] - th0[0]}, throttled time +{(th1[1] - th0[1]) / 1e3:.1f} ms", flush=True)
    t, h = np.array(st["time"][2:]) * 1e3, np.array(st["host_time"][2:]) * 1e3
    worst = np.argsort(t)[::-1][:4]
    print(f"run {r}: {len(t) / t.sum() * 1e3:8.0f} frames/s, median {np.median(t):.3f} ms, host median {np.median(h):.3f} ms; slowest: " +
          ", ".join(f"frame {2 + i}: dev {t[i]:.2f} ms host {h[i]:.2f} ms" for i in worst), flush=True)
# ---- which call holds the host in the one-time stall of a process's first run?  (second process state: rerun in a fresh process with "first")
