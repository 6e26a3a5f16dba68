cd $GRAFT_REPO_ROOT
python -m pytest tests/test_torch_stream.py tests/test_gpu_ops.py tests/test_single_touch.py -m gpu -x -q 2>&1 | tail -3
python tools/bench_mt.py 2>&1 | tail -6
python tools/bench_score_mfma.py 2>&1 | tail -1
tools/prof_stats.sh r04_score_mfma 200 python tools/bench_score_one.py | grep -E "k_score_mfma|k_codes_prepare|TFLOP" | cut -c1-220
PMC_GROUPS=tools/pmc_groups_mfma.txt tools/pmc.sh r04_score_mfma "k_score_mfma" python tools/bench_score_one.py
python tools/pmc_summary.py gpurun_out/pmc_r04_score_mfma k_score_mfma > gpurun_out/r04_pmc_score_mfma.txt 2>&1; cat gpurun_out/r04_pmc_score_mfma.txt | head -60
