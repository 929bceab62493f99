#!/bin/bash
# pyrdown2 strip height sweep (debug-knob build, LKFLOW_PD2_TH): per-stage times of the headline step and of the 4K / 8K / webcam shapes
set -e
OUT=gpurun_out/${1:-ab_pd2}; mkdir -p $OUT
export LKFLOW_LIB=$PWD/optical-flow_amd/liblkflow_two.so
{
for th in 8 16 24 32 36 40 44 48 52 56 60 64 68 72 8 48; do
  echo -n "th=$th 1080p x32: "; LKFLOW_PD2_TH=$th timeout -k 10 120 python tools/stage_times.py 32 1080 1920 3 15 20 2>/dev/null | grep sum= | cut -d' ' -f1
done
for th in 8 32 48 64 96 136 8; do
  echo -n "th=$th 4k x8: "; LKFLOW_PD2_TH=$th timeout -k 10 120 python tools/stage_times.py 8 2160 3840 5 21 10 2>/dev/null | grep sum= | cut -d' ' -f1
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
