#!/bin/bash
# round-end evidence: GPU suite, default bench line, rocprofv3 kernel traces + PMC passes for every config (-> gpurun_out/prof_<tag>)
TAG=${1:-r04_e}
O=gpurun_out/final_$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
bash tools/prof_round.sh $TAG > $O/prof.log 2>&1; tail -3 $O/prof.log
cp gpurun_out/prof_$TAG/pmc_traffic.json profiles/pmc_traffic.json
S0=$SECONDS; timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall $((SECONDS - S0)) s"
python bench.py --steps 20 --profile-all --no-extras > $O/bench_stages.json 2>> $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_rocprof"), d["roofline"]["traffic"], d["path_roofline"])
for k in ("config_4k","config_webcam","config_8k","config_pacman"):
    c=d.get(k)
    if c: print(k, c["value"], c["ms_per_step"], c["roofline"]["frac"], c["roofline"].get("frac_rocprof"), c["roofline"]["traffic"], c["path_roofline"]["frac_of_8TBps"])
print(json.load(open("$O/bench_stages.json"))["stage_ms_per_step"])
PY
