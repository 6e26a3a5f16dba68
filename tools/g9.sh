cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_single_touch.py tests/test_cluster_centers.py -m gpu -x -q 2>&1 | tail -3
python tools/bench_score_mfma.py 2>&1 | tail -1
for v in noload noepi nothing pf4; do echo -n "$v: "; MIDAS_HIP_LIB=$PWD/midastouch_amd/csrc/build/variants/$v.so python tools/bench_score_one.py 2>&1 | tail -1; done
tools/prof_stats.sh r04_score_mfma 200 python tools/bench_score_one.py | grep -E "k_score_mfma|k_codes_prepare|TFLOP" | cut -c1-220
