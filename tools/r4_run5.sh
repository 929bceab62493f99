#!/bin/bash
O=gpurun_out/r4g; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py -q -m gpu -x -k "fused_warp or every_window or full_size or config" > $O/test.log 2>&1; echo "tests rc=$?"; tail -3 $O/test.log
for w in 23 25 27 29 31; do
  echo "== 4K L5 w$w fused:   $(FUSE=1 timeout -k 10 120 python tools/stage_times.py 8 2160 3840 5 $w | head -1)" | tee -a $O/stage.txt
  echo "== 4K L5 w$w unfused: $(FUSE=0 timeout -k 10 120 python tools/stage_times.py 8 2160 3840 5 $w | head -1)" | tee -a $O/stage.txt
done
for w in 5 9; do
  echo "== 1080p L3 w$w fused:   $(FUSE=1 timeout -k 10 120 python tools/stage_times.py 32 1080 1920 3 $w | head -1)" | tee -a $O/stage.txt
  echo "== 1080p L3 w$w unfused: $(FUSE=0 timeout -k 10 120 python tools/stage_times.py 32 1080 1920 3 $w | head -1)" | tee -a $O/stage.txt
done
