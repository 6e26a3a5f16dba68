#!/bin/bash
# usage: tools/pmc.sh <tag> <kernel-regex> <python script and args...>   (GPU box only; writes gpurun_out/pmc_<tag>/)
# PMC_GROUPS=<file>: one counter group per line instead of the default SQ groups.
# One rocprofv3 --pmc pass per counter group (kernel-trace only), restricted to kernels matching the regex.
set -u
TAG=$1; shift
RE=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --kernel-include-regex "$RE" --pmc $group -d $OUT/p$i -o p$i --output-format csv -- "$@" > $OUT/p$i.log 2>&1
  echo "pass $i ($group): rc=$?"
done < <(if [ -n "${PMC_GROUPS:-}" ]; then cat "$PMC_GROUPS"; else cat <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
GROUPS
fi)
