#!/bin/bash
# Build tuning variants of libmidas_hip.so:  tools/variants.sh name "-DMIDAS_NN_BATCH=16 ..." [name flags]...
# Outputs midastouch_amd/csrc/build/variants/<name>.so ; run one with MIDAS_HIP_LIB=<path>.
set -e
cd "$(dirname "$0")/../midastouch_amd/csrc"
make -s
mkdir -p build/variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc $F $flags -c score.hip -o build/variants/$name.score.o &&
    /opt/rocm/bin/hipcc $F $flags -c particles.hip -o build/variants/$name.particles.o &&
    /opt/rocm/bin/hipcc $F $flags -c resample.hip -o build/variants/$name.resample.o &&
    /opt/rocm/bin/hipcc $F $flags -c api.hip -o build/variants/$name.api.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/$name.so build/variants/$name.score.o build/variants/$name.particles.o build/variants/$name.resample.o build/variants/$name.api.o &&
    echo built $name ) &
done
wait
