cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config5" 2>&1 | tail -3
python -m pytest tests/test_gpu_pipelined.py -m gpu -x -q 2>&1 | tail -3
echo "--- c5 presort on (default)"; python tools/bench_c5.py 2>&1 | grep "c5 init"
echo "--- c5 presort off"; MIDAS_PRESORT=0 python tools/bench_c5.py 2>&1 | grep "c5 init"
echo "--- c2 presort on"; MIDAS_PRESORT=1 python bench.py --no-cpu-baseline --no-extras --no-loop 2>&1 | tail -1 | cut -c1-200
echo "--- c2 presort off"; python bench.py --no-cpu-baseline --no-extras --no-loop 2>&1 | tail -1 | cut -c1-200
tools/prof_stats.sh r04_c5 300 python tools/bench_c5.py | grep -E "k_frame_front|k_presort|k_tail" | cut -c1-200
