#!/bin/bash
# HBM traffic of the step kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (GPU box only).
# usage: tools/pmc_traffic.sh <tag>
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace --kernel-include-regex "k_frame_front|k_score_reg|k_particle_update|k_tail_a|k_tail_b" --pmc $c -d $OUT/$n -o p --output-format csv -- python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-profile --no-loop --no-extras > $OUT/$n.log 2>&1
  echo "$c rc=$?"
done
python tools/pmc_summary.py $OUT k_ > $OUT/summary.txt
cat $OUT/summary.txt
