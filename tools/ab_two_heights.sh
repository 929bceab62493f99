#!/bin/bash
# A/B of the two-strip-height launch plan (lkflow.hip pick_strip_plan) on one box: debug-knob build, LKFLOW_TWO_HEIGHTS=0 / 1.
# usage (GPU box): bash tools/ab_two_heights.sh <out dir under gpurun_out>
set -e
OUT=gpurun_out/${1:-ab_two}; mkdir -p $OUT
export LKFLOW_LIB=$PWD/optical-flow_amd/liblkflow_two.so
{
for rep in 1 2; do
for t in 0 1; do
  echo "== LKFLOW_TWO_HEIGHTS=$t (rep $rep)"
  echo -n "1080p L3 w15 x32:   "; LKFLOW_TWO_HEIGHTS=$t timeout -k 10 120 python tools/stage_times.py 32 1080 1920 3 15 20 | head -1
  echo -n "4k L5 w21 x8:       "; LKFLOW_TWO_HEIGHTS=$t timeout -k 10 120 python tools/stage_times.py 8 2160 3840 5 21 10 | head -1
  echo -n "240x160 L3 w15 x244:"; LKFLOW_TWO_HEIGHTS=$t timeout -k 10 120 python tools/stage_times.py 244 160 240 3 15 20 | head -1
  echo -n "8k L5 w31 x4:       "; LKFLOW_TWO_HEIGHTS=$t timeout -k 10 120 python tools/stage_times.py 4 4320 7680 5 31 5 | head -1
done; done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
