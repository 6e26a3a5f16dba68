#!/bin/bash
cd /root/repo
for v in default pipe; do
  if [ $v = default ]; then unset MIDAS_HIP_LIB; else export MIDAS_HIP_LIB=$PWD/midastouch_amd/csrc/build/variants/$v.so; fi
  echo -n "$v: "; python tools/bench_topn.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['k_selfsim_mfma_4096']['tflops'],1), round(d['k_selfsim_mfma_8192']['tflops'],1), 'e2e', round(d['panel_4096']['seconds']*1e3,2))"
done
