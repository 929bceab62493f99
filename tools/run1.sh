set -e
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3b
timeout -k 10 1150 python -m pytest tests -m gpu -x -q > gpurun_out/r3b/suite.log 2>&1 || { tail -60 gpurun_out/r3b/suite.log; exit 1; }
tail -3 gpurun_out/r3b/suite.log
