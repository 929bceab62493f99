#!/usr/bin/env python3
"""Experiment: split the batch over S HIP streams (one lkflow context each) so that the VALU-bound LK
kernels of one chunk can overlap the memory-bound pyrDown / warp kernels of another.
    python tools/streams_exp.py [pairs_total] [streams...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "optical-flow_amd"))
import torch, lkflow_amd as lkf, synth

P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
variants = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
rows, cols, levels, win = 1080, 1920, 3, 15
prev, nxt = synth.frame_pairs_torch(P, rows, cols, "cuda", seed=1)
U = torch.empty_like(prev); V = torch.empty_like(prev)
for S in variants:
    ctxs = [lkf.Context(0) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    chunk = P // S
    for c in ctxs: c.reserve(rows, cols, levels, chunk)
    def step():
        for k in range(S):
            sl = slice(k * chunk, (k + 1) * chunk)
            ctxs[k].compute_batch(prev[sl], nxt[sl], levels, win, out=(U[sl], V[sl]), stream=streams[k])
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"streams={S} chunk={chunk}: {dt*1e3:.3f} ms/step  {P*rows*cols/dt/1e6:.0f} Mpx/s")
    for c in ctxs: c.close()
