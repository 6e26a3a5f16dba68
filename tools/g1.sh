cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r04_a_gputests.log 2>&1; tail -5 gpurun_out/r04_a_gputests.log
MIDAS_SCRATCH_LOG=1 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_a_bench_driver_line.json 2> gpurun_out/r04_a_bench_driver.err; cut -c1-400 gpurun_out/r04_a_bench_driver_line.json; tail -20 gpurun_out/r04_a_bench_driver.err
