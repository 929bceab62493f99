#!/bin/bash
# A/B the LK kernel variants built by tools/build_variant.sh: tools/ab.sh A B C ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  LKFLOW_LIB=$R/optical-flow_amd/liblkflow_$v.so python3 $R/bench.py --steps 10 --profile-all --no-cpu-baseline 2>/dev/null | \
   python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
done
