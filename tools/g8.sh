cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_loop.py tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -4
for v in noprio noload nolds noepi nothing; do echo -n "$v: "; MIDAS_HIP_LIB=$PWD/midastouch_amd/csrc/build/variants/$v.so python tools/bench_score_one.py 2>&1 | tail -1; done
