# kernel-trace of one library variant: tools/fwprof.sh <variant> [bench args]
V=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export LKFLOW_LIB=$R/optical-flow_amd/liblkflow_$V.so
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/kt_$V
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$V -- python3 $R/bench.py --steps 10 --no-extras "$@" > /tmp/kt_$V.json 2> /tmp/kt_$V.log || tail -5 /tmp/kt_$V.log
t=$(find /tmp/kt_$V -name "*kernel_trace.csv" | head -1)
python3 $R/tools/kernel_trace_by_grid.py $t
