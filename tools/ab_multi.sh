#!/bin/bash
# A/B library variants over several configs: tools/ab_multi.sh A B  (alternating, 2 rounds per config)
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { LKFLOW_LIB=$R/optical-flow_amd/liblkflow_$1.so python3 $R/bench.py --steps 20 --profile-all --no-extras --no-cpu-baseline "${@:2}" 2>/dev/null | \
   python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('$*', d['value'], d['ms_per_step'], {k:v for k,v in s.items() if k.startswith('lk')})"; }
for cfg in "--config 1080p" "--config 1080p --pairs 16" "--config 1080p --pairs 64" "--config 4k" "--config 8k --steps 5" "--config pacman" "--config webcam"; do
  for round in 1 2; do for v in "$@"; do run $v $cfg; done; done
done
