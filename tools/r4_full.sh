#!/bin/bash
O=gpurun_out/${1:-r4f}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["path_roofline"])
for k in ("config_4k","config_webcam","config_8k","config_pacman"):
    c=d.get(k); 
    if c: print(k, c["value"], c["ms_per_step"], c["roofline"]["frac"], c["path_roofline"]["frac_of_8TBps"])
PY
