#!/bin/bash
# A/B one environment knob of the library, alternating, three rounds: tools/ab_env2.sh VAR val_a val_b [bench args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
VAR=$1; A=$2; B=$3; shift 3
for round in 1 2 3; do for v in $A $B; do
  env $VAR=$v python3 $R/bench.py --steps 20 --profile-all --no-extras --no-cpu-baseline "$@" 2>/dev/null | \
   python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
done; done
