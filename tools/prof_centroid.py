#!/usr/bin/env python3
"""Timing of the flow-consumer reduction (f1) and the uint8 conversion on 32 x 1080p, HIP events via torch."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "optical-flow_amd"))
import torch
import lkflow_amd as L

ctx = L.Context(0)
P, H, W = 32, 1080, 1920
U = torch.randn((P, H, W), dtype=torch.float64, device="cuda") * 15
V = torch.randn((P, H, W), dtype=torch.float64, device="cuda") * 15
u8 = torch.randint(0, 256, (P, H, W), dtype=torch.uint8, device="cuda")


def t(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


px = P * H * W
for name, fn, bpp in (("flow_centroid", lambda: ctx.flow_centroid(U, V, 20.0), 16), ("u8_to_f64", lambda: ctx.u8_to_f64(u8), 9)):
    ms = t(fn)
    print(f"{name:14s} {ms:.4f} ms  {bpp * px / ms / 1e6:.0f} GB/s algorithmic")
