#!/bin/bash
# coarsest levels of odd height (240 x 135): the wave-specialised kernel (product library) -- compare with section G's "before" lines
set -e
OUT=gpurun_out/${1:-ab_odd}; mkdir -p $OUT
run() { echo -n "$1: "; timeout -k 10 120 python tools/stage_times.py $2 2>/dev/null | grep sum=; }
{
  run "4k L5 w21 x8      " "8 2160 3840 5 21 10"
  run "960x540 L3 w7 x32 " "32 540 960 3 7 10"
  run "960x540 L3 w13 x32" "32 540 960 3 13 10"
  run "960x540 L3 w19 x32" "32 540 960 3 19 10"
  run "960x540 L3 w25 x32" "32 540 960 3 25 10"
  run "960x540 L3 w29 x32" "32 540 960 3 29 10"
  run "4k L5 w21 x1      " "1 2160 3840 5 21 10"
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
