# correctness + A/B of the fused-warp variant build: tools/fwtest.sh [variant]
V=${1:-fw}
set -e
export LKFLOW_LIB=$GRAFT_REPO_ROOT/optical-flow_amd/liblkflow_$V.so
timeout -k 10 120 python __graft_entry__.py smoke
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size_1080p or block_boundary_widths_against_oracle or (config_shapes and not 31 and not 21) or sharded or inplace" 2>&1 | tail -5
unset LKFLOW_LIB
bash tools/ab.sh base $V base $V
