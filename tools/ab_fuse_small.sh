#!/bin/bash
# The fuse rule's small-launch clause: LKFLOW_OPT_FUSE_WARP = 0 (never) / 2 (rule) / 1 (always) on small shapes, same box.
set -e
OUT=gpurun_out/${1:-ab_fuse_small}; mkdir -p $OUT
run() { echo -n "$1: "; FUSE=$f timeout -k 10 120 python tools/stage_times.py $2 2>/dev/null | grep sum=; }
{
for f in 0 2 1 0 2 1; do
  echo "== LKFLOW_OPT_FUSE_WARP=$f"
  run "640x448 L5 w7 x16  " "16 448 640 5 7 20"
  run "640x448 L5 w15 x1  " "1 448 640 5 15 20"
  run "240x160 L3 w15 x1  " "1 160 240 3 15 20"
  run "240x160 L3 w15 x16 " "16 160 240 3 15 20"
  run "240x160 L3 w15 x32 " "32 160 240 3 15 20"
  run "240x160 L3 w15 x64 " "64 160 240 3 15 20"
  run "240x160 L3 w15 x244" "244 160 240 3 15 20"
  run "480x272 L3 w15 x8  " "8 272 480 3 15 20"
  run "480x272 L3 w15 x64 " "64 272 480 3 15 20"
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
