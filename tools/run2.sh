set -e
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3a
python tools/dbg_case.py 13 1 > gpurun_out/r3a/dbg_rel_f1.txt 2>&1
python tools/dbg_case.py 13 0 > gpurun_out/r3a/dbg_rel_f0.txt 2>&1
export LKFLOW_LIB=$PWD/optical-flow_amd/liblkflow_dbg.so
LKFLOW_BATCH=0 LKFLOW_BATCH_LK=0 python tools/dbg_case.py 13 0 > gpurun_out/r3a/dbg_old.txt 2>&1
LKFLOW_BATCH=0 LKFLOW_BATCH_LK=1 python tools/dbg_case.py 13 0 > gpurun_out/r3a/dbg_lkonly.txt 2>&1
cat gpurun_out/r3a/dbg_*.txt
