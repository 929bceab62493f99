#!/bin/bash
# I-cache / fetch counters for bench.py kernels: bash tools/prof_icache.sh <tag> [bench args]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for P in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -- python3 $R/bench.py --no-extras "$@" > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
  i=$((i+1))
done
python3 $R/tools/pmc_summary.py $OUT | grep -A14 "lk_split_kernel<15, 2, true, false>"
