#!/bin/bash
# Evidence for profiles/: rocprofv3 kernel-trace stats and PMC passes of bench.py on the headline config and on every extra
# BASELINE shape (1080p, 4k, webcam, 8k, pacman).  Usage (GPU box, repo root):  bash tools/prof_round.sh <tag>      e.g. r04_a
# Output: gpurun_out/prof_<tag>/ (copy what should be judged into profiles/) incl. pmc_traffic.json (all workloads merged).
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
declare -A WL=( [1080p]=1920x1080x32 [4k]=3840x2160x8 [webcam]=1080x1080x48 [8k]=7680x4320x32 [pacman]=240x160x244 )
declare -A STEPS=( [1080p]=20 [4k]=10 [webcam]=20 [8k]=3 [pacman]=20 )
for cfg in ${CFGS:-1080p 4k webcam 8k pacman}; do
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$cfg -- python3 $R/bench.py --config $cfg --steps ${STEPS[$cfg]} --no-extras \
      > $OUT/${TAG}_${cfg}_bench_under_rocprof.json 2> $OUT/kt_$cfg.log || { echo "kernel-trace $cfg failed"; tail -3 $OUT/kt_$cfg.log; }
  f=$(find $OUT/kt_$cfg -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_${cfg}_kernel_stats.csv
  t=$(find $OUT/kt_$cfg -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python3 $R/tools/kernel_trace_by_grid.py $t > $OUT/${TAG}_${cfg}_kernel_trace_by_grid.csv && cat $OUT/${TAG}_${cfg}_kernel_trace_by_grid.csv
  rm -rf $OUT/kt_$cfg
  cd $R
  s=3; [ $cfg = 8k ] && s=1
  TRACE_BY_GRID=$OUT/${TAG}_${cfg}_kernel_trace_by_grid.csv WORKLOAD=${WL[$cfg]} bash tools/prof_pmc.sh ${TAG}_$cfg --config $cfg --steps $s --warmup 1 > /dev/null
  cp gpurun_out/pmc_${TAG}_$cfg.txt $OUT/${TAG}_${cfg}_pmc_summary.txt
  cp gpurun_out/pmc_${TAG}_${cfg}_traffic.json $OUT/${TAG}_${cfg}_pmc_traffic.json
  rm -rf gpurun_out/pmc_${TAG}_$cfg
done
python3 tools/merge_pmc_traffic.py $OUT/pmc_traffic.json $OUT/${TAG}_*_pmc_traffic.json
