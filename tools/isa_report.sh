#!/bin/bash
# ISA histogram of the inner loops of the hot kernels -> stdout (committed as profiles/r0N_*_isa_histogram.txt).  No GPU needed.
cd "$(dirname "$0")/../optical-flow_amd"
S=/tmp/lkflow_isa.s
hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -DLKFLOW_FAST_BUILD -S --cuda-device-only -o $S csrc/lkflow.hip 2>/dev/null || exit 1
T=../tools/isa_loops.py
echo "ISA histogram of the inner loops (tools/isa_loops.py on hipcc -S --cuda-device-only, -DLKFLOW_FAST_BUILD)."
echo "STATIC instruction counts per loop body by class.  The wave-uniform 'some tap leaves the image' branches (W role: 4 select trees of ~38"
echo "v_cndmask per 2-step body; remap_blend in warp_kernel / lk_w3x1f) are in the static count but rarely executed.  The S role's loop unrolls 2 WIN"
echo "pipeline steps (the longest entry is listed); B, F and W loops hold 2 steps (2 barriers)."
echo
echo "== fused lk_split_kernel<15, 2, true, true>: S (30 steps) | B (2 steps) | F (2 steps) | W, rows E = 0 and E = 1 (2 steps each) | frame pixels"
python3 $T $S _ZN2lk15lk_split_kernelILi15ELi2ELb1ELb1EEEvNS_6LKArgsE 300 | grep -Ev "s_barrier=([3-9]|1[0-9]|2[0-9]) "
echo
echo "== unfused lk_split_kernel<15, 2, true, false>: S | F (2 steps, global loads) | B"
python3 $T $S _ZN2lk15lk_split_kernelILi15ELi2ELb1ELb0EEEvNS_6LKArgsE 200 | grep -Ev "s_barrier=([3-9]|1[0-9]|2[0-9]) "
echo
echo "== warp_kernel<true> (one row pair per body)"
python3 $T $S _ZN2lk11warp_kernelILb1EEEvNS_8WarpArgsE 100
echo
echo "== lk_w3x1f_kernel<true> (3 marched rows per body)"
python3 $T $S _ZN2lk15lk_w3x1f_kernelILb1EEEvNS_6LKArgsE 1000
echo
echo "== pyrdown2_kernel and pyrdown_kernel (2 output rows per body)"
python3 $T $S _ZN2lk15pyrdown2_kernelENS_12PyrDown2ArgsE 300
python3 $T $S _ZN2lk14pyrdown_kernelENS_11PyrDownArgsE 200
