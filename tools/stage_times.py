"""Per-stage HIP-event times of one pyramid step and of a plain single-level LK launch, for A/B runs of kernel
variants (LKFLOW_LIB + the debug-knob build).  usage: stage_times.py [pairs] [rows] [cols] [levels] [win] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "optical-flow_amd"))
import torch
import lkflow_amd as lk
import synth
P, H, W, L, win, reps = (int(x) for x in (sys.argv[1:7] + ["32", "1080", "1920", "3", "15", "12"][len(sys.argv) - 1:]))
prev, nxt = synth.frame_pairs_torch(P, H, W, "cuda", seed=1234)
ctx = lk.Context(0)
fuse = os.environ.get("FUSE")
if fuse is not None: ctx.set_option(lk.LKFLOW_OPT_FUSE_WARP, int(fuse))
tile = os.environ.get("TILE")
if tile is not None: ctx.set_option(lk.LKFLOW_OPT_TILE_ROWS, int(tile))
U = torch.empty_like(prev); V = torch.empty_like(prev)
for _ in range(40): ctx.compute_batch(prev, nxt, L, win, out=(U, V))
torch.cuda.synchronize()
ctx.profile()
for _ in range(reps): ctx.compute_batch(prev, nxt, L, win, out=(U, V))
torch.cuda.synchronize()
pr = ctx.profile_read()
names = {0: "pyrdown", 1: "warp", 2: "lk"}
tot = 0.0
out = []
for (st, lv), (ms, n) in sorted(pr.items()):
    out.append(f"{names[st]}{lv}={ms / reps:.3f}")
    tot += ms / reps
print(" ".join(out), f"| sum={tot:.3f} ms/step  {P * H * W / tot / 1e3:.0f} Mpx/s")
# plain single-level LK (no coarser level)
for _ in range(5): ctx.lk_batch(prev, nxt, win)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): ctx.lk_batch(prev, nxt, win)
e1.record(); torch.cuda.synchronize()
print(f"single-level LK: {e0.elapsed_time(e1) / reps:.3f} ms")
