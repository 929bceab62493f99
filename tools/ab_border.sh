R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { # name lib env...
  n=$1; lib=$2; shift 2
  env "$@" LKFLOW_LIB=$R/optical-flow_amd/$lib python3 $R/bench.py --steps 20 --profile-all --no-extras --no-cpu-baseline 2>/dev/null | \
   python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
}
for round in 1 2 3; do
  run base liblkflow_base.so LKFLOW_BORDER_SIDE=0
  run merged1 liblkflow_merged.so LKFLOW_BORDER_MERGED=1
  run merged0 liblkflow_merged.so LKFLOW_BORDER_MERGED=0
done
