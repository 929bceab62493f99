#!/bin/bash
# Build an A/B variant of the library: tools/build_variant.sh <name> [extra hipcc flags...]
# -> optical-flow_amd/liblkflow_<name>.so (select at run time with LKFLOW_LIB=<path>).
set -e
NAME=$1; shift
cd "$(dirname "$0")/../optical-flow_amd"
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -ffp-contract=off -DLKFLOW_FAST_BUILD "$@" \
  -Rpass-analysis=kernel-resource-usage -o liblkflow_$NAME.so csrc/lkflow.hip 2> /tmp/build_$NAME.log || { grep error -A3 /tmp/build_$NAME.log | head -20; exit 1; }
grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy" /tmp/build_$NAME.log | paste - - - - - | sed 's/remark: //g; s/csrc\/[a-z_.0-9]*:[0-9]*:[0-9]*://g; s/\[-Rpass-analysis=kernel-resource-usage\]//g' | awk '{$1=$1};1' | grep "march3.*Li15E" | sed 's/Function Name: _ZN2lk16lk_march3_kernelI/  /'
