#!/bin/bash
# A/B library variants on another bench config: tools/ab_cfg.sh <config> A B C ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$1; shift
for v in "$@"; do
  LKFLOW_LIB=$R/optical-flow_amd/liblkflow_$v.so python3 $R/bench.py --config $C --steps 10 --profile-all --no-extras 2>/dev/null | \
   python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], {k:v for k,v in d['stage_ms_per_step'].items() if k.startswith('lk')})"
done
