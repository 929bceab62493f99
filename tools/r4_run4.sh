#!/bin/bash
O=gpurun_out/r4e; mkdir -p $O
V=$1
export LKFLOW_LIB=$PWD/optical-flow_amd/liblkflow_$V.so
timeout -k 10 300 python -m pytest tests/test_gpu_round2.py -q -m gpu -k "fused_warp" -x > $O/${V}_test.log 2>&1; echo "$V fw test rc=$?"; tail -2 $O/${V}_test.log
for cfg in "32 1080 1920 3 15" "8 2160 3840 5 21" "244 160 240 3 15" "8 2160 3840 3 15"; do
  echo "== $cfg fused"; FUSE=1 timeout -k 10 120 python tools/stage_times.py $cfg | head -1 | tee -a $O/${V}_stage.txt
  echo "== $cfg unfused"; FUSE=0 timeout -k 10 120 python tools/stage_times.py $cfg | head -1 | tee -a $O/${V}_stage.txt
done
