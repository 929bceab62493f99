#!/bin/bash
# A/B runs of library variants and knobs on the GPU box (folds the round-2 one-liners ab_border / ab_cfg / ab_cfg2 / ab_env /
# ab_env2 / ab_inflight / ab_multi / ab_split / ab_win into one script).  Variants are built by tools/build_variant.sh
# (-> optical-flow_amd/liblkflow_<name>.so, add -DLKFLOW_DEBUG_KNOBS for the environment knobs: the SHIPPED library reads
# no environment variable) and selected with LKFLOW_LIB by the Python binding.
#
#   tools/ab.sh [-c config] [-w win] [-s steps] [-r rounds] [-x "extra bench args"] ITEM...
#     ITEM = variant            liblkflow_<variant>.so with the default knobs      (variant "-" = the shipped library)
#          | variant:VAR=V,...  the same with environment knobs (needs a -DLKFLOW_DEBUG_KNOBS variant)
#   prints per item: Mpixels/s, ms per step and the per-stage HIP-event times; rounds > 1 alternates the items.
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=1080p; WIN=0; STEPS=10; ROUNDS=1; EXTRA=""
while getopts "c:w:s:r:x:" o; do
  case $o in c) CFG=$OPTARG;; w) WIN=$OPTARG;; s) STEPS=$OPTARG;; r) ROUNDS=$OPTARG;; x) EXTRA=$OPTARG;; *) exit 2;; esac
done
shift $((OPTIND - 1))
for round in $(seq $ROUNDS); do
  for item in "$@"; do
    v=${item%%:*}; knobs=""
    [ "$item" != "$v" ] && knobs=$(echo "${item#*:}" | tr ',' ' ')
    lib=""; [ "$v" != "-" ] && lib="LKFLOW_LIB=$R/optical-flow_amd/liblkflow_$v.so"
    env $knobs $lib python3 $R/bench.py --config $CFG --win $WIN --steps $STEPS --profile-all --no-extras --no-cpu-baseline $EXTRA 2>/dev/null | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$item]', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
  done
done
