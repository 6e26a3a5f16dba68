#!/bin/bash
# A/B of the pipelined front on the bench workload (GPU box): parity tests of the engines first, then bench.py per setting.
# usage: tools/ab_front.sh <tag> [name ENV=val,ENV=val ...]   (variant libraries under midastouch_amd/csrc/build/variants/*.so are included)
cd "$(dirname "$0")/.."
TAG=${1:-ab}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ -z "$NOTESTS" ]; then
python -m pytest tests/test_gpu_pipelined.py tests/test_gpu_step.py tests/test_gpu_goldens_r2.py tests/test_gpu_loop.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
fi
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-loop > $OUT/$name.json 2> $OUT/$name.err
  python - "$name" "$OUT/$name.json" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["value"]), "steps/s", {k: round(v * 1e3, 1) for k, v in d["roofline"]["per_kernel_ms"].items()}, "median", round(d["config"]["per_step"]["ms_per_step_median"], 4))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
run default X=1
while [ $# -ge 2 ]; do n=$1; e=$(echo $2 | tr ',' ' '); shift 2; run $n $e; done
for f in midastouch_amd/csrc/build/variants/*.so; do
  [ -f "$f" ] || continue
  n=$(basename $f .so)
  run $n MIDAS_HIP_LIB=$PWD/$f
done
