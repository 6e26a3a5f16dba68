python -X faulthandler - <<'PY' 2>&1 | tail -40
import torch, os, sys
sys.path.insert(0, os.getcwd())
from midastouch_amd import _lib
ctx = _lib.context()
print("ctx ok", flush=True)
import numpy as np
from midastouch_amd.engine import PipelinedFilterEngine
from midastouch_amd.synthetic import make_codebook, make_trajectory
cb = make_codebook("004_sugar_box", K=2000, D=128, seed=1)
tr = make_trajectory(cb, T=6, seed=2)
dev = torch.device("cuda", 0)
eng = PipelinedFilterEngine(cb.poses, cb.embeddings, cb.mesh_vertices, 5000, seed=3, device=dev)
print("engine ok", flush=True)
eng.set_particles(torch.as_tensor(cb.poses[np.random.default_rng(0).integers(0, 2000, 5000)]))
eng.project_to_codebook()
print("proj ok", flush=True)
od, co = torch.as_tensor(tr.odoms).to(dev), torch.as_tensor(tr.codes).to(dev)
for t in range(1, 5):
    eng.step(od[t], co[t]); torch.cuda.synchronize(); print("step", t, flush=True)
PY
