set -e
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3c
python tools/_chk.py 17 19 21 23 25 27 29 31 2>&1 | grep -v amdgpu.ids
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3c/suite.log 2>&1 || { tail -40 gpurun_out/r3c/suite.log; exit 1; }
tail -2 gpurun_out/r3c/suite.log
python tools/stage_times.py 32 1080 1920 3 15 20 2>&1 | grep "sum="
python tools/stage_times.py 8 2160 3840 5 21 10 2>&1 | grep "sum="
