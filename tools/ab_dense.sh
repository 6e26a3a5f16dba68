#!/bin/bash
cd /root/repo
run() { name=$1; shift; env "$@" python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-loop 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$name', round(d['value']), 'steps/s', 'first', round(c['diffuse_regime']['ms_first_frame']*1e3,1), 'diffuse mean', round(c['diffuse_regime']['ms_per_step_mean']*1e3,1))"; }
for rep in 1 2; do
run off MIDAS_DENSE_ROWS=0
run d3125_w1024 X=1
run d3125_w2048 MIDAS_LIST_WAVES=2048
run d3125_w4096 MIDAS_LIST_WAVES=4096
run d1500_w4096 MIDAS_LIST_WAVES=4096 MIDAS_DENSE_ROWS=1500
run d6000_w4096 MIDAS_LIST_WAVES=4096 MIDAS_DENSE_ROWS=6000
done
