set -e
cd $GRAFT_REPO_ROOT
export LKFLOW_BATCH=1 FUSE=0
for v in dbg abl_NODERIV abl_NOSOLVE; do
  echo "== $v"; LKFLOW_LIB=$PWD/optical-flow_amd/liblkflow_$v.so python tools/stage_times.py 2>&1 | grep "single-level"
done
