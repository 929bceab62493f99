#!/bin/bash
# the 270 x 270 x 48 level (level 2 of 1080 x 1080 / L4): fused by plan / fused with one forced height / unfused
set -e
OUT=gpurun_out/${1:-ab_small}; mkdir -p $OUT
{
for rep in 1 2; do
echo -n "plan       FUSE=2: "; FUSE=2 timeout -k 10 120 python tools/stage_times.py 48 1080 1080 4 15 10 2>/dev/null | grep sum=
for t in 270 136 90 68 46; do
echo -n "TILE=$t FUSE=1: "; FUSE=1 TILE=$t timeout -k 10 120 python tools/stage_times.py 48 1080 1080 4 15 10 2>/dev/null | grep sum=
done
echo -n "plan       FUSE=0: "; FUSE=0 timeout -k 10 120 python tools/stage_times.py 48 1080 1080 4 15 10 2>/dev/null | grep sum=
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
