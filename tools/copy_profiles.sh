#!/bin/bash
# Copies the summaries of a tools/round_end.sh tag from gpurun_out/ into profiles/ (tracked) and refreshes the three files bench.py
# cites (rNN_traffic.json, rNN_traffic_dense.json, rNN_front_trace.json, rNN_dense_front_trace.json).  usage: tools/copy_profiles.sh <tag> <rNN>
cd "$(dirname "$0")/.."
T=$1; R=${2:-r06}
for f in gpurun_out/${T}_*; do
  case "$f" in *.err) continue;; esac
  [ -f "$f" ] && cp "$f" profiles/
done
for k in traffic traffic_dense front_trace dense_front_trace; do
  [ -f gpurun_out/${T}_$k.json ] && cp gpurun_out/${T}_$k.json profiles/${R}_$k.json
done
ls profiles | grep -c "^${T}_"
