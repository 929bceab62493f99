#!/usr/bin/env python3
"""Per-kernel timing of the f2 pre-processing chain (33 frames of 1080x1080 BGR), HIP events via torch."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "optical-flow_amd"))
import torch
import lkflow_amd as L

ctx = L.Context(0)
F, S = 33, 1080
bgr = torch.randint(0, 256, (F, S, S, 3), dtype=torch.uint8).cuda()
gray = ctx.blur_gray(bgr); diff = ctx.frame_diff(gray); edge = ctx.edge_map(diff)


def t(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


px = F * S * S
for name, fn, bpp in (("blur_gray", lambda: ctx.blur_gray(bgr), 4), ("frame_diff", lambda: ctx.frame_diff(gray), 3),
                      ("edge_map", lambda: ctx.edge_map(diff), 9), ("diff_edge_map", lambda: ctx.diff_edge_map(gray), 10)):
    ms = t(fn)
    print(f"{name:11s} {ms:.4f} ms  {bpp * px / ms / 1e6:.0f} GB/s algorithmic")
