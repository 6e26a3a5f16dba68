cd $GRAFT_REPO_ROOT
for r in 0 4 8 16 32; do echo "--- presort run=$r"; MIDAS_PRESORT=1 MIDAS_PRESORT_RUN=$r python tools/bench_c5.py 2>&1 | grep "c5 init"; done
echo "--- presort off"; python tools/bench_c5.py 2>&1 | grep "c5 init"
MIDAS_PRESORT=1 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config5" 2>&1 | tail -2
