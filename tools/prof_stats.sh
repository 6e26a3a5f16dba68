#!/bin/bash
# rocprofv3 --kernel-trace --stats of a command (GPU box only); the kernel_stats csv is copied to gpurun_out/<tag>_kernel_stats.csv
# usage: tools/prof_stats.sh <tag> <timeout_s> <command ...>
set -u
TAG=$1; LIMIT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
( cd "$GRAFT_REPO_ROOT" && timeout "$LIMIT" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o p -- "$@" ) > "$OUT/run.log" 2>&1
echo "rc=$?"
tail -3 "$OUT/run.log" | cut -c1-400
f=$(find "$OUT" -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv"; head -28 "$f" | cut -c1-160; else echo "no kernel_stats.csv under $OUT"; fi
