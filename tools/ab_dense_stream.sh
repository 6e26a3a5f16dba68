#!/bin/bash
# A/B of the DENSE front (all K rows streamed beside the particle waves, MIDAS_DENSE_SCORES=1) over the variant libraries under
# midastouch_amd/csrc/build/variants/*.so (GPU box): 200 steps, kernel times by HIP events.   usage: tools/ab_dense_stream.sh <tag>
cd "$(dirname "$0")/.."
TAG=${1:-abdense}; OUT=gpurun_out/$TAG; mkdir -p $OUT
one() { local name=$1 lib=$2
  env ${lib:+MIDAS_HIP_LIB=$lib} MIDAS_DENSE_SCORES=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-loop --no-diffuse --no-extras > $OUT/$name.json 2> $OUT/$name.err
  python - "$name" "$OUT/$name.json" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "dense c2", round(d["value"]), "steps/s", {k: round(v * 1e3, 1) for k, v in d["roofline"]["per_kernel_ms"].items()}, "frac", round(d["roofline"]["frac"], 3))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
for rep in 1 2; do
one default ""
for f in midastouch_amd/csrc/build/variants/*.so; do [ -f "$f" ] && one $(basename $f .so) $PWD/$f; done
done
