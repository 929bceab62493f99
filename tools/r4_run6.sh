#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py -q -m gpu -x -k "fused_warp or every_window or full_size or config or fuzz" > $O/test.log 2>&1; echo "tests rc=$?"; tail -3 $O/test.log
for P in 1 4; do
  echo "== 1080p P=$P fused:   $(FUSE=1 timeout -k 10 120 python tools/stage_times.py $P 1080 1920 3 15 | head -1)" | tee -a $O/stage.txt
  echo "== 1080p P=$P unfused: $(FUSE=0 timeout -k 10 120 python tools/stage_times.py $P 1080 1920 3 15 | head -1)" | tee -a $O/stage.txt
done
CFGS="1080p" bash tools/prof_round.sh r04_a > $O/prof.log 2>&1; tail -5 $O/prof.log
cat gpurun_out/prof_r04_a/r04_a_1080p_pmc_summary.txt | grep -A40 "lk_split_kernel<15, 2, true, true>" | head -60
