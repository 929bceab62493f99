#!/bin/bash
# A/B two environment settings of the default library: tools/ab_env.sh "VAR=1" "" "VAR=1" "" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for e in "$@"; do
  env $e python3 $R/bench.py --steps 10 --profile-all --no-extras 2>/dev/null | \
   python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$e]', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
done
