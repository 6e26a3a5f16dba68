#!/bin/bash
# usage: tools/pmc.sh <tag> <python script and args...>   (GPU box only; writes gpurun_out/pmc_<tag>/)
# One rocprofv3 --pmc pass per counter group (kernel-trace only, no other trace domains).
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $group -d $OUT/p$i -o p$i --output-format csv -- "$@" > $OUT/p$i.log 2>&1
  echo "pass $i ($group): rc=$?"
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_INSTS_BRANCH GRBM_GUI_ACTIVE
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
FETCH_SIZE
WRITE_SIZE
GROUPS
find $OUT -name "*.csv" | head -40
