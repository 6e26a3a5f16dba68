cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, '.')
from midastouch_amd.engine import FilterEngine, PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
dev = torch.device("cuda", 0)
def run(eng, step, n=100, warm=10):
    for i in range(warm): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): step(warm + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for T in (130, 262):
    cb = make_codebook(K=5000, D=256, seed=1000); tr = make_trajectory(cb, T=T, seed=2000)
    od, co = torch.as_tensor(tr.odoms).to(dev), torch.as_tensor(tr.codes).to(dev)
    for tag, cls in (("eager", FilterEngine), ("pipelined", PipelinedFilterEngine)):
        eng = cls(cb.poses, cb.embeddings, cb.mesh_vertices, 1000, device=dev)
        eng.set_particles(torch.as_tensor(cb.poses[np.random.default_rng(0).integers(0, 5000, 1000)])); eng.project_to_codebook()
        for rep in range(4):
            us = run(eng, lambda i: eng.step(od[1 + i % (T-2)], co[1 + i % (T-2)]))
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for i in range(50): eng.step(od[1 + i % (T-2)], co[1 + i % (T-2)])
            ev1.record(); torch.cuda.synchronize()
            print(T, tag, rep, "%.1f us/step wall, %.1f us/step device events" % (us, ev0.elapsed_time(ev1) * 20), "valid", int(eng.status[1]), "tele", eng.telemetry.cpu().numpy()[:4])
PY
