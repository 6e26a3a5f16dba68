for a in 0 1 2 3; do
  echo "ABLATE=$a"
  MIDAS_ABLATE=$a timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['per_kernel_ms'])"
done
