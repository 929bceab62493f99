#!/bin/bash
# Batch-size sweep of the headline config: tools/sweep_pairs.sh [config] -- per-stage HIP-event times per step and the
# time per pair, to see whether small batches (working set inside the 256 MB Infinity Cache) run faster per pair.
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=${1:-1080p}
for p in 1 2 4 8 16 32 64; do
  python3 $R/bench.py --config $C --pairs $p --steps 20 --profile-all --no-extras --no-cpu-baseline 2>/dev/null | \
   python3 -c "import sys,json; d=json.loads(sys.stdin.read()); p=$p; print('P=%d'%p, 'Mpx/s', d['value'], 'ms/step', d['ms_per_step'], 'us/pair', round(1e3*d['ms_per_step']/p,1), {k:round(v,4) for k,v in d['stage_ms_per_step'].items()})"
done
