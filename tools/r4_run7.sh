#!/bin/bash
O=gpurun_out/r4i; mkdir -p $O
export LKFLOW_LIB=$PWD/optical-flow_amd/liblkflow_fw10.so
timeout -k 10 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py -q -m gpu -x -k "solve or win3 or warp or edge_map_stream_1080 or golden or t1 or t2 or t3" > $O/test.log 2>&1; echo "tests rc=$?"; tail -3 $O/test.log
for V in fw10 fw10a fw10b fw10c fw10 fw10a fw10b; do
  export LKFLOW_LIB=$PWD/optical-flow_amd/liblkflow_$V.so
  echo "== $V $(FUSE=1 timeout -k 10 120 python tools/stage_times.py 2>/dev/null | head -1)" | tee -a $O/stage.txt
done
export LKFLOW_LIB=$PWD/optical-flow_amd/liblkflow_fw10.so
echo "== webcam $(timeout -k 10 120 python tools/stage_times.py 48 1080 1080 4 3 2>/dev/null| head -1)" | tee -a $O/stage.txt
echo "== unfused $(FUSE=0 timeout -k 10 120 python tools/stage_times.py 2>/dev/null | head -1)" | tee -a $O/stage.txt
