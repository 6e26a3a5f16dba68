#!/usr/bin/env python3
"""The seeded mode at c2 (bench.py's config.parity_mode) on its own, for the chunked / sequential generator and piece counts:
   python tools/bench_parity_mode.py [pieces ...]"""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from midastouch_amd.synthetic import make_codebook, make_trajectory
from midastouch_amd.tactile_tree import tactile_tree
from midastouch_amd import torch_rng
dev = torch.device("cuda", 0)
torch.set_num_threads(bench.cpu_quota())
N, K, D = 100_000, 50_000, 512
cb = make_codebook("004_sugar_box", K=K, D=D, seed=1001)
traj = make_trajectory(cb, T=202, seed=2001)
tree = tactile_tree(torch.as_tensor(cb.poses), torch.as_tensor(cb.cam_poses), torch.as_tensor(cb.embeddings))
tree.to_device(dev)
from midastouch_amd import ops
mesh = ops.Tree(torch.as_tensor(cb.mesh_vertices).to(dev, torch.float64))
orig = torch_rng.TorchCpuStream.__init__
for pieces in [int(a) for a in sys.argv[1:]] or [0, 4, 8, 16]:
    def init(self, seed, device=None, overlap=True, pieces=pieces, _o=orig):
        _o(self, seed, device, overlap, pieces)
    torch_rng.TorchCpuStream.__init__ = init
    r = bench.parity_mode_rate(cb, traj, N, dev, tree, mesh)
    print(f"pieces={pieces}: {r['steps_per_sec']:.0f} steps/s ({r['ms_per_step'] * 1e3:.1f} us/step)  runs {r['steps_per_sec_runs']}", flush=True)
    if os.environ.get("ALL_DRAWS"):  # every draw of the frame from the stream (motion noise too)
        r = bench.parity_mode_rate(cb, traj, N, dev, tree, mesh, motion=True)
        print(f"pieces={pieces}, all draws: {r['steps_per_sec']:.0f} steps/s ({r['ms_per_step'] * 1e3:.1f} us/step)  runs {r['steps_per_sec_runs']}", flush=True)
# the generator alone
st = torch_rng.TorchCpuStream.__new__(torch_rng.TorchCpuStream)
for pieces in (0, 4, 8, 16, 32):
    st = torch_rng.TorchCpuStream.__new__(torch_rng.TorchCpuStream)
    orig(st, 3000, dev, True, pieces)
    out = torch.empty(N, dtype=torch.float64, device=dev)
    for _ in range(3):
        st.rand64(N, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        st.rand64_async(N, out)
    torch.cuda.synchronize()
    print(f"generator alone, pieces={pieces}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call of N={N}", flush=True)
