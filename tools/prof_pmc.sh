#!/bin/bash
# Collect rocprofv3 PMC counters for bench.py in separate passes (counters only with --kernel-trace).
# Usage (on the GPU box, from the repo root):  bash tools/prof_pmc.sh <tag> [bench args...]
# Output: gpurun_out/pmc_<tag>/<pass>/..._counter_collection.csv  +  gpurun_out/pmc_<tag>.txt summary
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY"
 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
 "GRBM_GUI_ACTIVE"
)
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -- python3 $R/bench.py --no-extras "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
  i=$((i+1))
done
python3 $R/tools/pmc_summary.py $OUT $R/gpurun_out/pmc_${TAG}_traffic.json ${WORKLOAD:-1920x1080x32} ${TRACE_BY_GRID:-} > $R/gpurun_out/pmc_$TAG.txt 2>&1
cat $R/gpurun_out/pmc_$TAG.txt
