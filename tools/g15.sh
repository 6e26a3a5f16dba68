cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_pipelined.py tests/test_gpu_step.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | head
python tools/bench_c5.py 2>&1 | grep "c5 init"
MIDAS_PRESORT_FUSED=0 python tools/bench_c5.py 2>&1 | grep "c5 init"
tools/prof_stats.sh r04_c5c 300 python tools/bench_c5.py > /dev/null; grep -E "k_frame_front|k_presort|k_tail|k_rmse" gpurun_out/r04_c5c_kernel_stats.csv | sed 's/(.*)"//' | cut -c1-150
