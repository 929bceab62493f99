R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in base idle0; do for f in 1 2 4; do
  LKFLOW_LIB=$R/optical-flow_amd/liblkflow_$v.so python3 $R/bench.py --steps 10 --no-extras --inflight $f 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v inflight=$f', d['value'], d['ms_per_step'])"
done; done
