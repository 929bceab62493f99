#!/bin/bash
# calibration of the strip cost model (rounds x (th + overhead)): forced single strip heights (LKFLOW_OPT_TILE_ROWS) on 32 x 1080p
set -e
OUT=gpurun_out/${1:-ab_model}; mkdir -p $OUT
{
for f in 1 0; do
for t in 1080 540 360 270 216 180 136 120 90 68 54 46; do
  echo -n "FUSE=$f TILE=$t: "; FUSE=$f TILE=$t timeout -k 10 120 python tools/stage_times.py 32 1080 1920 3 15 12 2>/dev/null | grep sum= | cut -d'|' -f1
done; done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
