#!/bin/bash
# Warp fused into the LK kernel by the shipped rule (LKFLOW_OPT_FUSE_WARP = 2) against never (0) and always (1), per level, same box.
# usage (GPU box): bash tools/ab_fuse_levels.sh <out dir under gpurun_out>
set -e
OUT=gpurun_out/${1:-ab_fuse}; mkdir -p $OUT
run() { echo -n "$1: "; FUSE=$f timeout -k 10 120 python tools/stage_times.py $2 2>/dev/null | grep sum=; }
{
for f in 0 2 1 0 2 1; do
  echo "== LKFLOW_OPT_FUSE_WARP=$f"
  run "1080p L3 w15 x32   " "32 1080 1920 3 15 20"
  run "1080p L3 w15 x4    " "4 1080 1920 3 15 20"
  run "1080p L3 w15 x1    " "1 1080 1920 3 15 20"
  run "960x540 L3 w15 x32 " "32 540 960 3 15 20"
  run "960x540 L3 w15 x2  " "2 540 960 3 15 20"
  run "4k L5 w21 x8       " "8 2160 3840 5 21 10"
  run "4k L5 w21 x1       " "1 2160 3840 5 21 10"
  run "480x272 L3 w15 x64 " "64 272 480 3 15 20"
  run "640x448 L5 w7 x16  " "16 448 640 5 7 20"
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
