#!/bin/bash
# A/B of the batch step (c5) and the c2 bench over the variant libraries under midastouch_amd/csrc/build/variants/*.so (GPU box).
# usage: tools/ab_c5.sh <tag>
cd "$(dirname "$0")/.."
TAG=${1:-abc5}; OUT=gpurun_out/$TAG; mkdir -p $OUT
one() { # name, lib or ""
  local name=$1 lib=$2
  echo "== $name"
  env ${lib:+MIDAS_HIP_LIB=$lib} timeout 300 python tools/bench_c5.py 2>&1 | grep "^c5" | tee -a $OUT/$name.c5.log
  if [ -z "$NOC2" ]; then
  env ${lib:+MIDAS_HIP_LIB=$lib} timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-loop --no-diffuse > $OUT/$name.json 2> $OUT/$name.err
  python - "$name" "$OUT/$name.json" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "c2", round(d["value"]), "steps/s", {k: round(v * 1e3, 1) for k, v in d["roofline"]["per_kernel_ms"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
  fi
}
one default ""
for f in midastouch_amd/csrc/build/variants/*.so; do
  [ -f "$f" ] || continue
  one $(basename $f .so) $PWD/$f
done
