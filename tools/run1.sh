set -e
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3a
python tools/dbg_case.py 13 1 2>&1 | grep -v amdgpu.ids
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/suite.log 2>&1 || { tail -40 gpurun_out/r3a/suite.log; exit 1; }
tail -3 gpurun_out/r3a/suite.log
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err || { tail -20 gpurun_out/r3a/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3a/bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','roofline','path_roofline','stage_ms')})
PY
