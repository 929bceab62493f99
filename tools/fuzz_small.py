#!/usr/bin/env python3
"""One-off fuzz: many random small (rows, cols, levels, window, batch) cases of the HIP path against the oracle."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "optical-flow_amd")); sys.path.insert(0, R)
import torch
import lkflow_amd as L
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = L.Context(0)
wins = [3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31]
worst = 0.0
for it in range(n):
    lv_target = int(rng.integers(1, 6))
    h = int(rng.integers(1, 48)) << (lv_target - 1)
    w = int(rng.integers(1, 48)) << (lv_target - 1)
    levels = min(lv_target, L.getMaxLayer(w, h))
    win = wins[int(rng.integers(0, len(wins)))]
    P = int(rng.integers(1, 4))
    a = rng.random((P, h, w)); b = rng.random((P, h, w))
    if rng.random() < 0.3:
        a[0, :, : w // 2] = 0.5; b[0, :, : w // 2] = 0.5
    video = rng.random() < 0.3 and P > 1
    if video:
        fr = torch.from_numpy(np.concatenate([a, b[-1:]])).cuda()
        U, V = ctx.compute_video(fr, levels, win)
        pa, pb = fr[:-1].cpu().numpy(), fr[1:].cpu().numpy()
    else:
        U, V = ctx.compute_batch(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), levels, win)
        pa, pb = a, b
    U, V = U.cpu().numpy(), V.cpu().numpy()
    for p in range(P):
        uo, vo = O.coarse_to_fine(pa[p], pb[p], levels, win)
        scale = np.maximum(1.0, np.maximum(np.abs(uo), np.abs(vo)))
        e = max((np.abs(U[p] - uo) / scale).max(), (np.abs(V[p] - vo) / scale).max())
        worst = max(worst, e)
        if not e <= 1e-6:
            print("MISMATCH", it, h, w, levels, win, P, video, e); sys.exit(1)
print(f"fuzz ok: {n} cases, worst relative error {worst:.3e}")
