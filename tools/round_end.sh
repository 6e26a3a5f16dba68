#!/bin/bash
# End-of-round artefacts (GPU box): GPU test log, the bench line with the DRIVER'S flags (--gpus 1 --steps 20 --warmup 5: what
# BENCH_rNN records) and with 200 steps, rocprofv3 kernel statistics of both, per-launch statistics of the driver's timed region,
# HBM traffic (PMC passes), other configs, the reference-named loop.  usage: tools/round_end.sh <tag>   -> gpurun_out/<tag>_*
cd "$(dirname "$0")/.."
T=${1:-rXX}
python -m pytest tests -m gpu -q > gpurun_out/${T}_gputests.log 2>&1; grep -aE "passed|failed" gpurun_out/${T}_gputests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver_line.json 2> gpurun_out/${T}_bench_driver.err; cut -c1-300 gpurun_out/${T}_bench_driver_line.json
python bench.py --no-cpu-baseline --no-extras > gpurun_out/${T}_bench_line_200.json 2> gpurun_out/${T}_bench_200.err; cut -c1-300 gpurun_out/${T}_bench_line_200.json
tools/prof_stats.sh ${T}_driver 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-loop | head -8
f=$(find gpurun_out/prof_${T}_driver -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/driver_trace_stats.py $f 5 20 gpurun_out/${T}_driver_trace.json | head -30
tools/prof_stats.sh ${T}_steady 300 python bench.py --steps 200 --warmup 100 --no-cpu-baseline --no-loop --no-diffuse --no-extras | head -8
f=$(find gpurun_out/prof_${T}_steady -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/front_trace_stats.py $f 200 gpurun_out/${T}_front_trace.json
tools/pmc_traffic.sh ${T}_traffic > gpurun_out/${T}_pmc_traffic_summary.txt 2>&1; python tools/make_traffic_json.py gpurun_out/pmc_${T}_traffic gpurun_out/${T}_traffic.json | head -20
tools/prof_stats.sh ${T}_c5 300 python tools/bench_c5.py | head -8
python tools/bench_configs.py > gpurun_out/${T}_other_configs.json 2> gpurun_out/${T}_other_configs.err; cat gpurun_out/${T}_other_configs.json | cut -c1-600
python tools/bench_filter_loop.py > gpurun_out/${T}_filter_loop.jsonl 2>&1; tail -2 gpurun_out/${T}_filter_loop.jsonl | cut -c1-400
# round 4: PMC of the front (c2), of the batch front (c5) and of the batched scorer on the matrix cores
tools/pmc.sh ${T}_front "k_frame_front" python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-profile --no-loop --no-extras --no-diffuse > /dev/null 2>&1
PMC_GROUPS=tools/pmc_groups_mem.txt tools/pmc.sh ${T}_front_mem "k_frame_front" python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-profile --no-loop --no-extras --no-diffuse > /dev/null 2>&1
( python tools/pmc_summary.py gpurun_out/pmc_${T}_front k_frame_front; python tools/pmc_summary.py gpurun_out/pmc_${T}_front_mem k_frame_front ) > gpurun_out/${T}_pmc_front.txt 2>&1; head -20 gpurun_out/${T}_pmc_front.txt
tools/pmc.sh ${T}_c5 "k_frame_front" python tools/bench_c5.py > /dev/null 2>&1
PMC_GROUPS=tools/pmc_groups_mem.txt tools/pmc.sh ${T}_c5_mem "k_frame_front" python tools/bench_c5.py > /dev/null 2>&1
( python tools/pmc_summary.py gpurun_out/pmc_${T}_c5 k_frame_front; python tools/pmc_summary.py gpurun_out/pmc_${T}_c5_mem k_frame_front ) > gpurun_out/${T}_pmc_c5_front.txt 2>&1; head -8 gpurun_out/${T}_pmc_c5_front.txt
tools/prof_stats.sh ${T}_score_mfma 200 python tools/bench_score_one.py | grep -E "k_score_mfma|TFLOP" | cut -c1-200
PMC_GROUPS=tools/pmc_groups_mfma.txt tools/pmc.sh ${T}_score_mfma "k_score_mfma" python tools/bench_score_one.py > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${T}_score_mfma k_score_mfma > gpurun_out/${T}_pmc_score_mfma.txt 2>&1; head -20 gpurun_out/${T}_pmc_score_mfma.txt
python tools/bench_score_mfma.py > gpurun_out/${T}_score_mfma_shapes.json 2>/dev/null; cat gpurun_out/${T}_score_mfma_shapes.json | cut -c1-400
