#!/bin/bash
# tools/ab_cfg2.sh <config> lib1 lib2 ...   (A/B library variants on one bench config)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cfg=$1; shift
for v in "$@"; do
  LKFLOW_LIB=$R/optical-flow_amd/liblkflow_$v.so python3 $R/bench.py --config $cfg --steps 10 --profile-all --no-cpu-baseline --no-extras 2>/dev/null | \
   python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg $v', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
done
