#!/usr/bin/env python3
"""The one-time host stall in the FIRST run of the reference-named loop in a process (tools/diag_loop_stall.py: frame ~20, ~35 ms):
times LoopEngine.step and the C-ABI call inside it per frame, with MIDAS_SCRATCH_LOG=1."""
import gc, os, sys, time
os.environ["MIDAS_SCRATCH_LOG"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midastouch_amd import loop_engine, _lib
from midastouch_amd.config import load_config
from midastouch_amd.filter import Sequence, filter as run_filter
from midastouch_amd.synthetic import make_codebook, make_trajectory
from midastouch_amd.tactile_tree import tactile_tree
dev = torch.device("cuda", 0)
N, K, D, T = 100_000, 50_000, 512, 60
cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
traj = make_trajectory(cb, T=T + 2, seed=2001)
tree = tactile_tree(torch.as_tensor(cb.poses), torch.as_tensor(cb.cam_poses), torch.as_tensor(cb.embeddings))
tree.to_device(dev)
cfg = load_config([f"expt.params.num_particles={N}", f"expt.codebook_size={K}", f"tcn.model.output_dim={D}"])
seq = Sequence(torch.as_tensor(traj.gt_poses[:T]).to(dev), torch.as_tensor(traj.meas_poses[:T]).to(dev), torch.as_tensor(traj.codes[:T]).to(dev), tree,
               cb.mesh_vertices, "004_sugar_box")
lib = _lib.load()
t_step, t_c = [], []
orig_step = loop_engine.LoopEngine.step
orig_c = lib.midas_loop_step
class Timed:
    def __init__(self, f): self.f = f
    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.f(*a); t_c.append(time.perf_counter() - t0); return r
def step(self, *a, **kw):
    if not isinstance(self.ctx.lib.midas_loop_step, Timed):
        pass
    t0 = time.perf_counter(); r = orig_step(self, *a, **kw); t_step.append(time.perf_counter() - t0); return r
loop_engine.LoopEngine.step = step
class LibProxy:
    def __init__(self, lib): self._lib = lib; self.midas_loop_step = Timed(lib.midas_loop_step)
    def __getattr__(self, n): return getattr(self._lib, n)
ctx = _lib.context(dev)
ctx.lib = LibProxy(ctx.lib)
gc.collect(); gc.disable()
st = run_filter(cfg, seq, device=dev, floor=1000)
ts, tc, th = np.array(t_step) * 1e3, np.array(t_c) * 1e3, np.array(st["host_time"]) * 1e3
for i in np.argsort(th)[::-1][:5]:
    print(f"frame {i}: iteration {th[i]:.2f} ms, LoopEngine.step {ts[i]:.2f} ms, midas_loop_step (C) {tc[i]:.2f} ms, particles {st['num_particles'][i]}, mode {st['frames'][i]['mode']}")
