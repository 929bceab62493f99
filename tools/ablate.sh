#!/bin/bash
# timing-only ablations of the fused kernel (results wrong by design): tools/ablate.sh variant...
O=gpurun_out/ablate; mkdir -p $O
for V in "$@"; do
  export LKFLOW_LIB=$PWD/optical-flow_amd/liblkflow_$V.so
  echo "== $V $(FUSE=1 timeout -k 10 120 python tools/stage_times.py | head -1)" | tee -a $O/abl.txt
done
