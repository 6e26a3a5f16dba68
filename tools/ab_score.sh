#!/bin/bash
# A/B builds of score.hip only (the other objects are reused): tools/ab_score.sh name "-DFLAGS" [name flags]... (CPU box: build;
# GPU box: MIDAS_HIP_LIB=midastouch_amd/csrc/build/variants/<name>.so python tools/bench_score_one.py)
set -e
cd "$(dirname "$0")/../midastouch_amd/csrc"
make -s
mkdir -p build/variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc $F $flags -c score.hip -o build/variants/$name.score.o
    objs="build/variants/$name.score.o"
    for f in particles resample cluster topn selfsim loop topk_aten dbscan dbscan_nd index_build mt19937 comm api; do objs="$objs build/$f.o"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/$name.so $objs -ldl && echo built $name ) &
done
wait
