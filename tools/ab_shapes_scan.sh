#!/bin/bash
# per-level times on shapes whose coarse levels fall to the one-role kernel (odd widths, tiny levels), product library
set -e
OUT=gpurun_out/${1:-ab_scan}; mkdir -p $OUT
run() { echo -n "$1: "; timeout -k 10 120 python tools/stage_times.py $2 2>/dev/null | grep sum=; }
{
  run "1080x1080 L4 w15 x48  " "48 1080 1080 4 15 10"
  run "1080x1080 L4 w21 x48  " "48 1080 1080 4 21 10"
  run "1080x1080 L3 w15 x48  " "48 1080 1080 3 15 10"
  run "1000x1000 L4 w15 x32  " "32 1000 1000 4 15 10"
  run "1000x1000 L3 w15 x32  " "32 1000 1000 3 15 10"
  run "640x448 L7 w9 x32     " "32 448 640 7 9 10"
  run "640x448 L5 w15 x32    " "32 448 640 5 15 10"
  run "3264x2448 L3 w15 x4   " "4 2448 3264 3 15 10"
  run "3264x2448 L5 w15 x4   " "4 2448 3264 5 15 10"
  run "1280x720 L5 w15 x32   " "32 720 1280 5 15 10"
  run "1280x720 L4 w21 x32   " "32 720 1280 4 21 10"
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
