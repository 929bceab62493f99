#!/bin/bash
# Sweep the strip heights of the pyrDown / warp kernels (debug env knobs): tools/sweep_th.sh <variant>
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=${1:-cur}
run() { LKFLOW_LIB=$R/optical-flow_amd/liblkflow_$V.so python3 $R/bench.py --steps 10 --profile-all --no-cpu-baseline 2>/dev/null | \
   python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('$1', d['ms_per_step'], s['pyrdown_L1'], s['pyrdown_L2'], s['warp_L0'], s['warp_L1'])"; }
for t in 8 12 16 17 18 20 24 27 30 34 45 54 68 90; do LKFLOW_PD_TH=$t run "pd$t"; done
for t in 8 12 16 20 24 30 36 40 54 60 72 90 108 136 180; do LKFLOW_WARP_TH=$t run "warp$t"; done
