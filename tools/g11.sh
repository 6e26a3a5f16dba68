cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-loop 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver', round(d['value']), d['roofline']['per_kernel_ms'])"
python bench.py --no-cpu-baseline --no-extras --no-loop 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('200', round(d['value']), d['roofline']['per_kernel_ms'])"
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config5 or config2" 2>&1 | tail -2
MIDAS_PRESORT=1 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config5" 2>&1 | tail -2
echo "--- ablate 4 (stats)"; MIDAS_PIPELINED=1 MIDAS_ABLATE=4 python tools/scan_stats.py 2>&1 | grep -E "ticks per wave|mean wave lifetime"
echo "--- ablate 12 (no search)"; MIDAS_PIPELINED=1 MIDAS_ABLATE=12 python tools/scan_stats.py 2>&1 | grep -E "ticks per wave|mean wave lifetime"
echo "--- ablate 5 (trust hint)"; MIDAS_PIPELINED=1 MIDAS_ABLATE=5 python tools/scan_stats.py 2>&1 | grep -E "ticks per wave|mean wave lifetime"
echo "--- ablate 6 (no prune)"; MIDAS_PIPELINED=1 MIDAS_ABLATE=6 python tools/scan_stats.py 2>&1 | grep -E "ticks per wave|mean wave lifetime"
echo "--- ablate 15 (nothing)"; MIDAS_PIPELINED=1 MIDAS_ABLATE=15 python tools/scan_stats.py 2>&1 | grep -E "ticks per wave|mean wave lifetime"
