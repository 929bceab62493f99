#!/bin/bash
# pyrdown_kernel strip height on small levels (debug-knob build, LKFLOW_PD_TH)
set -e
OUT=gpurun_out/${1:-ab_pd}; mkdir -p $OUT
export LKFLOW_LIB=$PWD/optical-flow_amd/liblkflow_two.so
{
for th in 16 2 4 6 8 10 12 16 4 8; do
  echo -n "th=$th 240x160 x244: "; LKFLOW_PD_TH=$th timeout -k 10 120 python tools/stage_times.py 244 160 240 3 15 30 2>/dev/null | grep sum=
done
for th in 16 4 8 16 4 8; do
  echo -n "th=$th 4k x8: "; LKFLOW_PD_TH=$th timeout -k 10 120 python tools/stage_times.py 8 2160 3840 5 21 10 2>/dev/null | grep sum= | cut -d' ' -f1-4
  echo -n "th=$th 640x448 L5 x16: "; LKFLOW_PD_TH=$th timeout -k 10 120 python tools/stage_times.py 16 448 640 5 7 20 2>/dev/null | grep sum= | cut -d' ' -f1-5
  echo -n "th=$th 1080p x1: "; LKFLOW_PD_TH=$th timeout -k 10 120 python tools/stage_times.py 1 1080 1920 3 15 20 2>/dev/null | grep sum= | cut -d' ' -f1-3
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
