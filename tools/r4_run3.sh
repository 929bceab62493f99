#!/bin/bash
O=gpurun_out/r4c; mkdir -p $O
for V in "$@"; do
  export LKFLOW_LIB=$PWD/optical-flow_amd/liblkflow_$V.so
  timeout -k 10 200 python -m pytest tests/test_gpu_round2.py -q -m gpu -k "fused_warp" -x > $O/${V}_test.log 2>&1; echo "$V fw test rc=$?"; tail -1 $O/${V}_test.log
  echo "== $V"; FUSE=1 timeout -k 10 120 python tools/stage_times.py | tee -a $O/${V}_stage.txt
done
