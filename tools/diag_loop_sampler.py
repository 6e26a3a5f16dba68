#!/usr/bin/env python3
"""Which host call holds the one-time 23 / 43 ms pause in a process's first run of the reference-named loop?  A sampler thread records
the main thread's Python stack every millisecond; the samples that fall inside the slowest iteration are printed."""
import gc, os, sys, threading, time, traceback
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd.config import load_config
from midastouch_amd.filter import Sequence, filter as run_filter
from midastouch_amd.synthetic import make_codebook, make_trajectory
from midastouch_amd.tactile_tree import tactile_tree
dev = torch.device("cuda", 0)
N, K, D, T = 100_000, 50_000, 512, 120
cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
traj = make_trajectory(cb, T=T + 2, seed=2001)
tree = tactile_tree(torch.as_tensor(cb.poses), torch.as_tensor(cb.cam_poses), torch.as_tensor(cb.embeddings))
tree.to_device(dev)
cfg = load_config([f"expt.params.num_particles={N}", f"expt.codebook_size={K}", f"tcn.model.output_dim={D}"])
seq = Sequence(torch.as_tensor(traj.gt_poses[:T]).to(dev), torch.as_tensor(traj.meas_poses[:T]).to(dev), torch.as_tensor(traj.codes[:T]).to(dev), tree,
               cb.mesh_vertices, "004_sugar_box")
main_id = threading.main_thread().ident
samples, stop = [], False
def sampler():
    while not stop:
        f = sys._current_frames().get(main_id)
        if f is not None:
            st = traceback.extract_stack(f, limit=4)
            samples.append((time.perf_counter(), " <- ".join(f"{os.path.basename(s.filename)}:{s.lineno}({s.name})" for s in reversed(st))))
        time.sleep(0.001)
th = threading.Thread(target=sampler, daemon=True); th.start()
gc.collect(); gc.disable()
t_begin = time.perf_counter()
st = run_filter(cfg, seq, device=dev, floor=1000)
stop = True
h = np.array(st["host_time"])
i = int(np.argmax(h[2:])) + 2
print(f"slowest iteration: frame {i}, {h[i] * 1e3:.2f} ms host")
# iteration i's window: the host times are consecutive
t0 = t_begin
gaps = [(samples[j + 1][0] - samples[j][0], samples[j][1], samples[j + 1][1]) for j in range(len(samples) - 1)]
gaps.sort(reverse=True)
print("largest gaps between two samples of the sampler thread (a gap = nobody could take the GIL, or the sampler was not scheduled):")
for g, a, b in gaps[:3]:
    print(f"  {g * 1e3:.1f} ms between [{a}] and [{b}]")
# samples per location during the longest stretch where the main thread sat at one place
runs, cur = [], None
for t, loc in samples:
    if cur and cur[2] == loc:
        cur[1] = t
    else:
        cur = [t, t, loc]; runs.append(cur)
runs.sort(key=lambda r: r[0] - r[1])
for r in runs[:4]:
    print(f"  main thread stayed {1e3 * (r[1] - r[0]):.1f} ms at {r[2]}")
