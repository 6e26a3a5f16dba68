cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-loop 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver', round(d['value']), d['roofline']['per_kernel_ms'])"
python bench.py --no-cpu-baseline --no-extras --no-loop 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('200', round(d['value']), d['roofline']['per_kernel_ms'])"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-loop 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver', round(d['value']), d['roofline']['per_kernel_ms'])"
python -m pytest tests/test_gpu_step.py tests/test_gpu_pipelined.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -2
