cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config5" 2>&1 | tail -2
python -m pytest tests/test_gpu_pipelined.py tests/test_gpu_step.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
echo "--- presort on (default)"; python tools/bench_c5.py 2>&1 | grep "c5 init"
echo "--- presort off"; MIDAS_PRESORT=0 python tools/bench_c5.py 2>&1 | grep "c5 init"
tools/prof_stats.sh r04_c5b 300 python tools/bench_c5.py | grep -E "k_frame_front|k_presort|k_tail" | cut -c1-60,190-300
