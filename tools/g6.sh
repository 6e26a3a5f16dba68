cd $GRAFT_REPO_ROOT
python -m pytest tests/test_torch_stream.py tests/test_gpu_pipelined.py tests/test_gpu_step.py tests/test_gpu_ops.py tests/test_single_touch.py -m gpu -x -q 2>&1 | tail -5
python tools/bench_mt.py 2>&1 | tail -6
python tools/bench_score_mfma.py 2>&1 | tail -2
python tools/diag_c1.py 2>&1 | tail -14
python - <<'PY'
import sys, json, torch, numpy as np
sys.path.insert(0, '.')
import bench
from midastouch_amd.synthetic import make_codebook, make_trajectory
from midastouch_amd.tactile_tree import tactile_tree
from midastouch_amd import ops
dev = torch.device("cuda", 0)
cb = make_codebook("004_sugar_box", K=50000, D=512, seed=1001); traj = make_trajectory(cb, T=202, seed=2001)
tree = tactile_tree(torch.as_tensor(cb.poses), torch.as_tensor(cb.cam_poses), torch.as_tensor(cb.embeddings)); tree.to_device(dev)
mt = ops.Tree(torch.as_tensor(cb.mesh_vertices).to(dev, torch.float64))
print(json.dumps(bench.parity_mode_rate(cb, traj, 100000, dev, tree, mt)))
print(json.dumps(bench.config1_rates(dev)))
PY
tools/prof_stats.sh scoremfma 200 python tools/bench_score_mfma.py | grep -E "k_score_mfma|k_codes_prepare" | cut -c1-200
