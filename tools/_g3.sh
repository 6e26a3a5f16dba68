run() { python bench.py --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('steps/s %.0f  us/step %.2f  front %.2f  tail %.2f' % (d['value'], d['ms_per_step']*1e3, d['roofline']['per_kernel_ms']['frame_front']*1e3, d['roofline']['per_kernel_ms']['tail_a']*1e3))
"; }
for v in u8; do MIDAS_HIP_LIB=$PWD/midastouch_amd/csrc/build/variants/$v.so timeout 900 python -m pytest tests/test_gpu_pipelined.py -m gpu -q -x 2>&1 | tail -1; done
for rep in 1 2; do
echo "--- u16"; run
for v in u8; do echo "--- $v"; MIDAS_HIP_LIB=$PWD/midastouch_amd/csrc/build/variants/$v.so run; done
done
echo "--- off"; MIDAS_GUIDE=0 run
