#!/bin/bash
# run bench.py (no cpu baseline) for every variant library under build/variants, plus the default
cd "$(dirname "$0")/.."
run() { MIDAS_HIP_LIB=$2 timeout 300 python bench.py --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', round(d['value']), {k: round(v*1e3,1) for k,v in d[\"roofline\"][\"per_kernel_ms\"].items()})"; }
run default ""
for f in midastouch_amd/csrc/build/variants/*.so; do run $(basename $f .so) $PWD/$f; done
