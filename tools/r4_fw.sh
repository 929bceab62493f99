#!/bin/bash
# A/B of a fused-warp library variant on the GPU box: tools/r4_fw.sh <variant> [outdir]
V=$1; O=gpurun_out/${2:-r4fw}; mkdir -p $O
export LKFLOW_LIB=$PWD/optical-flow_amd/liblkflow_$V.so
timeout -k 10 300 python -m pytest tests/test_gpu_round2.py -q -m gpu -k "fused_warp" -x > $O/${V}_test.log 2>&1; echo "test rc=$?"; tail -3 $O/${V}_test.log
for r in 1 2; do
FUSE=1 timeout -k 10 120 python tools/stage_times.py | tee -a $O/${V}_stage.txt
FUSE=0 timeout -k 10 120 python tools/stage_times.py | tee -a $O/${V}_stage.txt
done
