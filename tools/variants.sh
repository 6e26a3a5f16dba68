#!/bin/bash
# Build tuning variants of libmidas_hip.so:  tools/variants.sh name "-DMIDAS_NN_BATCH=16 ..." [name flags]...
# Outputs midastouch_amd/csrc/build/variants/<name>.so ; run one with MIDAS_HIP_LIB=<path>.
# VFILES="particles" restricts the flags (and the recompilation) to the named translation units.
set -e
cd "$(dirname "$0")/../midastouch_amd/csrc"
make -s
mkdir -p build/variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( objs=""
    for f in score particles resample cluster topn selfsim loop topk_aten dbscan dbscan_nd index_build mt19937 comm api; do
      # VFILES="particles resample": only these units see the flags, the others are taken from the regular build
      if [ -n "$VFILES" ] && ! echo " $VFILES " | grep -q " $f "; then objs="$objs build/$f.o"; continue; fi
      /opt/rocm/bin/hipcc $F $flags -c $f.hip -o build/variants/$name.$f.o || exit 1
      objs="$objs build/variants/$name.$f.o"
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/$name.so $objs &&
    echo built $name ) &
done
wait
