#!/bin/bash
# End-of-round artefacts (GPU box): GPU test log, bench line, rocprofv3 kernel statistics (steady state, default bench, c5), per-launch
# trace statistics of the front, other configs, the reference-named loop.  usage: tools/round_end.sh <tag>   -> gpurun_out/<tag>_*
cd "$(dirname "$0")/.."
T=${1:-rXX}
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputests.log 2>&1; grep -aE "passed|failed" gpurun_out/${T}_gputests.log
python bench.py > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench.err; cut -c1-300 gpurun_out/${T}_bench_line.json
tools/prof_stats.sh ${T}_steady 300 python bench.py --steps 200 --warmup 100 --no-cpu-baseline --no-loop --no-diffuse | head -8
f=$(find gpurun_out/prof_${T}_steady -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/front_trace_stats.py $f 200 gpurun_out/${T}_front_trace.json
tools/prof_stats.sh ${T}_bench 400 python bench.py --no-cpu-baseline | head -6
tools/prof_stats.sh ${T}_c5 300 python tools/bench_c5.py | head -8
python tools/bench_configs.py > gpurun_out/${T}_other_configs.json 2> gpurun_out/${T}_other_configs.err; cat gpurun_out/${T}_other_configs.json | cut -c1-600
python tools/bench_filter_loop.py > gpurun_out/${T}_filter_loop.jsonl 2>&1; tail -2 gpurun_out/${T}_filter_loop.jsonl | cut -c1-400
