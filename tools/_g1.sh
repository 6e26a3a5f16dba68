timeout 900 python -m pytest tests/test_gpu_pipelined.py tests/test_gpu_step.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "config2" 2>&1 | tail -3
for g in 1 0; do echo "--- MIDAS_GUIDE=$g"; MIDAS_GUIDE=$g timeout 300 python bench.py --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('steps/s', d['value'], 'ms/step', d['ms_per_step'], 'front', d['roofline'].get('kernel_us') or d['roofline'])
"; done
