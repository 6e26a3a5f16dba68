#!/bin/bash
# End-of-round artefacts (GPU box): GPU test log, the bench line with the DRIVER'S flags (--gpus 1 --steps 20 --warmup 5: what
# BENCH_rNN records) and with 200 steps, rocprofv3 kernel statistics of both, per-launch statistics of the driver's timed region,
# HBM traffic (PMC passes) of the sparse AND the dense front, the tail, c3, c5 and the batched scorer on the matrix cores.
# usage: tools/round_end.sh <tag>   -> gpurun_out/<tag>_*      (copy what is to be judged into profiles/)
cd "$(dirname "$0")/.."
T=${1:-rXX}
B200="python bench.py --steps 200 --warmup 100 --no-cpu-baseline --no-loop --no-diffuse --no-extras"
B30="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-profile --no-loop --no-extras --no-diffuse"
python -m pytest tests -m gpu -q > gpurun_out/${T}_gputests.log 2>&1; grep -aE "passed|failed" gpurun_out/${T}_gputests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver_line.json 2> gpurun_out/${T}_bench_driver.err; cut -c1-300 gpurun_out/${T}_bench_driver_line.json
python bench.py --no-cpu-baseline --no-extras > gpurun_out/${T}_bench_line_200.json 2> gpurun_out/${T}_bench_200.err; cut -c1-300 gpurun_out/${T}_bench_line_200.json
# kernel traces: the driver's command (per-launch statistics of its timed region), the steady state, the dense front (all K rows streamed)
tools/prof_stats.sh ${T}_driver 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-loop | head -8
f=$(find gpurun_out/prof_${T}_driver -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/driver_trace_stats.py $f 5 20 gpurun_out/${T}_driver_trace.json | grep -E "mean_us|per_frame"
tools/prof_stats.sh ${T}_steady 300 $B200 | head -8
f=$(find gpurun_out/prof_${T}_steady -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/front_trace_stats.py $f 200 gpurun_out/${T}_front_trace.json | grep -E "mean_us_timed|\"frame|\"tail"
MIDAS_DENSE_SCORES=1 tools/prof_stats.sh ${T}_dense_steady 300 $B200 | head -8
f=$(find gpurun_out/prof_${T}_dense_steady -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/front_trace_stats.py $f 200 gpurun_out/${T}_dense_front_trace.json | grep -E "mean_us_timed|\"frame|\"tail"
# HBM traffic per launch (FETCH_SIZE / WRITE_SIZE in their own passes): sparse and dense
tools/pmc_traffic.sh ${T}_traffic > gpurun_out/${T}_pmc_traffic_summary.txt 2>&1; python tools/make_traffic_json.py gpurun_out/pmc_${T}_traffic gpurun_out/${T}_traffic.json | head -30
MIDAS_DENSE_SCORES=1 tools/pmc_traffic.sh ${T}_traffic_dense > gpurun_out/${T}_pmc_traffic_dense_summary.txt 2>&1; python tools/make_traffic_json.py gpurun_out/pmc_${T}_traffic_dense gpurun_out/${T}_traffic_dense.json | head -12
# SQ / memory-path counters: the sparse front, the dense front, the tail
for v in sparse dense; do
  if [ $v = dense ]; then export MIDAS_DENSE_SCORES=1; else unset MIDAS_DENSE_SCORES; fi
  tools/pmc.sh ${T}_front_$v "k_frame_front" $B30 > /dev/null 2>&1
  PMC_GROUPS=tools/pmc_groups_mem.txt tools/pmc.sh ${T}_front_${v}_mem "k_frame_front" $B30 > /dev/null 2>&1
  ( python tools/pmc_summary.py gpurun_out/pmc_${T}_front_$v k_frame_front; python tools/pmc_summary.py gpurun_out/pmc_${T}_front_${v}_mem k_frame_front ) > gpurun_out/${T}_pmc_front_$v.txt 2>&1
done
unset MIDAS_DENSE_SCORES
head -12 gpurun_out/${T}_pmc_front_sparse.txt
tools/pmc.sh ${T}_tail "k_tail_a3" $B30 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${T}_tail k_tail_a3 > gpurun_out/${T}_pmc_tail.txt 2>&1; head -12 gpurun_out/${T}_pmc_tail.txt
# c3 (one million particles on one GPU): kernel trace + counters
tools/prof_stats.sh ${T}_c3 300 python tools/bench_c3_one.py | head -6
tools/pmc.sh ${T}_c3 "k_frame_front" python tools/bench_c3_one.py > /dev/null 2>&1
PMC_GROUPS=tools/pmc_groups_mem.txt tools/pmc.sh ${T}_c3_mem "k_frame_front" python tools/bench_c3_one.py > /dev/null 2>&1
( python tools/pmc_summary.py gpurun_out/pmc_${T}_c3 k_frame_front; python tools/pmc_summary.py gpurun_out/pmc_${T}_c3_mem k_frame_front ) > gpurun_out/${T}_pmc_c3_front.txt 2>&1; head -8 gpurun_out/${T}_pmc_c3_front.txt
# c5 (64 trajectories x 10k particles): kernel trace + counters
tools/prof_stats.sh ${T}_c5 300 python tools/bench_c5.py | head -8
tools/pmc.sh ${T}_c5 "k_frame_front" python tools/bench_c5.py > /dev/null 2>&1
PMC_GROUPS=tools/pmc_groups_mem.txt tools/pmc.sh ${T}_c5_mem "k_frame_front" python tools/bench_c5.py > /dev/null 2>&1
( python tools/pmc_summary.py gpurun_out/pmc_${T}_c5 k_frame_front; python tools/pmc_summary.py gpurun_out/pmc_${T}_c5_mem k_frame_front ) > gpurun_out/${T}_pmc_c5_front.txt 2>&1; head -8 gpurun_out/${T}_pmc_c5_front.txt
# the batched scorer on the matrix cores
tools/prof_stats.sh ${T}_score_mfma 200 python tools/bench_score_one.py | grep -E "k_score_mfma|k_codes|TFLOP" | cut -c1-200
PMC_GROUPS=tools/pmc_groups_mfma.txt tools/pmc.sh ${T}_score_mfma "k_score_mfma" python tools/bench_score_one.py > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_${T}_score_mfma k_score_mfma > gpurun_out/${T}_pmc_score_mfma.txt 2>&1; head -20 gpurun_out/${T}_pmc_score_mfma.txt
python tools/bench_score_mfma.py > gpurun_out/${T}_score_mfma_shapes.json 2>/dev/null; cat gpurun_out/${T}_score_mfma_shapes.json | cut -c1-400
python tools/bench_configs.py > gpurun_out/${T}_other_configs.json 2> gpurun_out/${T}_other_configs.err; cat gpurun_out/${T}_other_configs.json | cut -c1-400
python tools/bench_filter_loop.py > gpurun_out/${T}_filter_loop.jsonl 2>&1; tail -2 gpurun_out/${T}_filter_loop.jsonl | cut -c1-300
# round 6: the seeded mode (device replica of torch's CPU generator, pieces side by side), the generator's kernels, the loop's and DBSCAN's kernels
python tools/bench_parity_mode.py 0 6 > gpurun_out/${T}_parity_mode.txt 2>&1; grep pieces gpurun_out/${T}_parity_mode.txt
tools/prof_stats.sh ${T}_mt19937 120 python tools/bench_mt_pieces.py 0 6 | grep -E "k_mt|rc="
tools/prof_stats.sh ${T}_loop 300 python tools/prof_loop.py 100000 1000 300 | grep -E "k_loop|k_front_small|rc=" | cut -c1-200
tools/prof_stats.sh ${T}_dbscan 300 python tools/prof_dbscan_frames.py | grep -E "k_db_|rc=" | cut -c1-160; grep -i slowest gpurun_out/prof_${T}_dbscan/run.log
# the raw traces are large: only the summaries travel back
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -size +2M -delete
