#!/bin/bash
# A/B the wave-specialised LK kernel against the one-role kernel on one box: tools/ab_split.sh [config...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "${@:-1080p}"; do
  for sp in 0 1; do
    LKFLOW_SPLIT=$sp python3 $R/bench.py --config $cfg --steps 10 --profile-all --no-cpu-baseline 2>/dev/null | \
     python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg split=$sp', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
  done
done
