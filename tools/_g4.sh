for v in u8 u8b2k; do
export MIDAS_HIP_LIB=$PWD/midastouch_amd/csrc/build/variants/$v.so
timeout 900 python -m pytest tests/test_gpu_pipelined.py -m gpu -q -x 2>&1 | tail -1
for g in 1; do
MIDAS_GUIDE=$g tools/prof_stats.sh guide$g 300 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-loop --no-diffuse --no-extras --no-profile 2>&1 | grep -E "rc=" | cut -c1-200
grep -E "k_frame_front<float, 8, 2|k_tail_a2d" gpurun_out/guide${g}_kernel_stats.csv | sed 's/"[^"]*"/K/' | cut -c1-120
done
done
