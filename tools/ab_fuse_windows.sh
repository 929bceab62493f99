#!/bin/bash
# every window 5..31 on 32 x 1080p / L3 and 8 x 4K / L4: warp fused by the shipped rule (2) against never (0), product library
set -e
OUT=gpurun_out/${1:-ab_wins}; mkdir -p $OUT
{
for w in 5 7 9 11 13 15 17 19 21 23 25 27 29 31; do
for f in 0 2; do
  echo -n "w=$w FUSE=$f 1080p x32 L3: "; FUSE=$f timeout -k 10 120 python tools/stage_times.py 32 1080 1920 3 $w 10 2>/dev/null | grep sum=
done; done
for w in 7 13 19 25 29; do
for f in 0 2; do
  echo -n "w=$w FUSE=$f 960x540 x32 L3: "; FUSE=$f timeout -k 10 120 python tools/stage_times.py 32 540 960 3 $w 10 2>/dev/null | grep sum=
done; done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
