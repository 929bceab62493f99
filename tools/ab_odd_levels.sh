#!/bin/bash
# coarsest levels of odd height / width (240 x 135, 135 x 135, 125 x 125) in the wave-specialised kernel (product library) -- compare
# with the "before" lines of profiles/r04_g_two_heights_ab.txt G
set -e
OUT=gpurun_out/${1:-ab_odd}; mkdir -p $OUT
run() { echo -n "$1: "; timeout -k 10 120 python tools/stage_times.py $2 2>/dev/null | grep sum=; }
{
  run "4k L5 w21 x8         " "8 2160 3840 5 21 10"
  run "960x540 L3 w25 x32   " "32 540 960 3 25 10"
  run "1080x1080 L4 w15 x48 " "48 1080 1080 4 15 10"
  run "1080x1080 L4 w21 x48 " "48 1080 1080 4 21 10"
  run "1080x1080 L4 w29 x48 " "48 1080 1080 4 29 10"
  run "1000x1000 L4 w15 x32 " "32 1000 1000 4 15 10"
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
