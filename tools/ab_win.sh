#!/bin/bash
# A/B library variants on a config with the window overridden: tools/ab_win.sh <config> <win> A B ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$1; W=$2; shift 2
for v in "$@"; do
  LKFLOW_LIB=$R/optical-flow_amd/liblkflow_$v.so python3 $R/bench.py --config $C --win $W --steps 10 --profile-all --no-extras 2>/dev/null | \
   python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v w$W', d['value'], d['ms_per_step'], {k:v for k,v in d['stage_ms_per_step'].items() if k.startswith('lk')})"
done
