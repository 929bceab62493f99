#!/bin/bash
# Evidence for profiles/: rocprofv3 kernel-trace stats and PMC passes of bench.py on the headline config and the
# extra BASELINE shapes.  Usage (GPU box, repo root):  bash tools/prof_round.sh <tag>      e.g. r02_a
# Output: gpurun_out/prof_<tag>/  (copy what should be judged into profiles/).
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for cfg in 1080p 4k webcam; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$cfg -- python3 $R/bench.py --config $cfg --steps 20 --no-extras \
      > $OUT/${TAG}_${cfg}_bench_under_rocprof.json 2> $OUT/kt_$cfg.log || { echo "kernel-trace $cfg failed"; tail -3 $OUT/kt_$cfg.log; }
  f=$(find $OUT/kt_$cfg -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_${cfg}_kernel_stats.csv
  t=$(find $OUT/kt_$cfg -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python3 $R/tools/kernel_trace_by_grid.py $t > $OUT/${TAG}_${cfg}_kernel_trace_by_grid.csv && cat $OUT/${TAG}_${cfg}_kernel_trace_by_grid.csv
  rm -rf $OUT/kt_$cfg
done
cd $R
TRACE_BY_GRID=$OUT/${TAG}_1080p_kernel_trace_by_grid.csv WORKLOAD=1920x1080x32 bash tools/prof_pmc.sh ${TAG}_1080p --steps 3 --warmup 1 > /dev/null
TRACE_BY_GRID=$OUT/${TAG}_4k_kernel_trace_by_grid.csv WORKLOAD=3840x2160x8 bash tools/prof_pmc.sh ${TAG}_4k --config 4k --steps 3 --warmup 1 > /dev/null
TRACE_BY_GRID=$OUT/${TAG}_webcam_kernel_trace_by_grid.csv WORKLOAD=1080x1080x48 bash tools/prof_pmc.sh ${TAG}_webcam --config webcam --steps 3 --warmup 1 > /dev/null
for cfg in 1080p 4k webcam; do
  cp gpurun_out/pmc_${TAG}_$cfg.txt $OUT/${TAG}_${cfg}_pmc_summary.txt
  cp gpurun_out/pmc_${TAG}_${cfg}_traffic.json $OUT/${TAG}_${cfg}_pmc_traffic.json
  rm -rf gpurun_out/pmc_${TAG}_$cfg
done
grep -A18 "lk_split_kernel<15, 2, true, false>" $OUT/${TAG}_1080p_pmc_summary.txt | head -30
