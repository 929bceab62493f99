set -e
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3a
export LKFLOW_LIB=$PWD/optical-flow_amd/liblkflow_dbg.so
export LKFLOW_BATCH=1
python tools/dbg_case.py 13 1 2>&1 | grep -v amdgpu.ids | tail -3
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config_shapes or full_size_1080p or zero_motion or batch_equals or video_batch_shares or strided" 2>&1 | tail -5
echo "batch fused:";    python tools/stage_times.py 2>&1 | grep -v amdgpu.ids
echo "batch unfused:";  FUSE=0 python tools/stage_times.py 2>&1 | grep -v amdgpu.ids
echo "old:";            LKFLOW_BATCH=0 FUSE=0 python tools/stage_times.py 2>&1 | grep -v amdgpu.ids
