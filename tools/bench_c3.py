import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from midastouch_amd.engine import FilterEngine, PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
def run(eng, step, n=40, warm=10):
    for i in range(warm): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): step(warm + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
cb = make_codebook("035_power_drill", K=50000, D=512, seed=1003); tr = make_trajectory(cb, T=70, seed=2003)
od, co = torch.as_tensor(tr.odoms).to(dev), torch.as_tensor(tr.codes).to(dev)
res = {}
for N in (1_000_000, 300_000):
    for tag, cls in (("eager", FilterEngine), ("pipelined", PipelinedFilterEngine)):
        eng = cls(cb.poses, cb.embeddings, cb.mesh_vertices, N, device=dev)
        eng.set_particles(torch.as_tensor(cb.poses[np.random.default_rng(2).integers(0, 50000, N)])); eng.project_to_codebook()
        res[f"N{N}_{tag}"] = round(run(eng, lambda i: eng.step(od[1 + i % 68], co[1 + i % 68])), 1)
        del eng
print(json.dumps(res))
