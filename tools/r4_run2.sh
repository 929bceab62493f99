#!/bin/bash
O=gpurun_out/r4b; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_round4.py tests/test_cpp_dropin.py -q -m gpu -x -s > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
for V in fw5 fw5pw2 fw5pf0 fw5pb2; do
  export LKFLOW_LIB=$PWD/optical-flow_amd/liblkflow_$V.so
  if [ $V = fw5 ]; then timeout -k 10 200 python -m pytest tests/test_gpu_round2.py -q -m gpu -k "fused_warp" -x > $O/${V}_test.log 2>&1; echo "fw test rc=$?"; tail -2 $O/${V}_test.log; fi
  echo "== $V"; FUSE=1 timeout -k 10 120 python tools/stage_times.py | tee -a $O/${V}_stage.txt
done
export LKFLOW_LIB=$PWD/optical-flow_amd/liblkflow_fw5.so
echo "== fw5 no border (timing only)"; LKFLOW_ABL_NO_BORDER=1 FUSE=1 timeout -k 10 120 python tools/stage_times.py
echo "== fw5 unfused"; FUSE=0 timeout -k 10 120 python tools/stage_times.py
unset LKFLOW_LIB
echo "== default lib"; timeout -k 10 120 python tools/stage_times.py; timeout -k 10 120 python tools/stage_times.py 48 1080 1080 4 3
