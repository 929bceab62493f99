"""Strip-height sweep of the LK launches (LKFLOW_OPT_TILE_ROWS): does the whole-rounds model of pick_tile_rows pick the
fastest strip height for the wave-specialised kernel?   python tools/sweep_lk_th.py [pairs] [config]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "optical-flow_amd"))
import torch
import lkflow_amd as lkf
import synth
import bench

P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
name = sys.argv[2] if len(sys.argv) > 2 else "1080p"
cols, rows, levels, win, _ = bench.CONFIGS[name]
dev = torch.device("cuda", 0)
prev, nxt = synth.frame_pairs_torch(P, rows, cols, dev, seed=1)
U, V = torch.empty_like(prev), torch.empty_like(prev)
ths = [0] + [t for t in (60, 72, 90, 108, 120, 136, 154, 180, 216, 270, 360, 540, 1080, 2160) if t <= rows]
with lkf.Context(0) as ctx:
    ctx.reserve(rows, cols, levels, P)
    stream = torch.cuda.current_stream()
    for th in ths + [0]:
        ctx.set_option(lkf.LKFLOW_OPT_TILE_ROWS, th)
        for _ in range(3):
            ctx.compute_batch(prev, nxt, levels, win, out=(U, V), stream=stream)
        torch.cuda.synchronize()
        ctx.profile()
        for _ in range(10):
            ctx.compute_batch(prev, nxt, levels, win, out=(U, V), stream=stream)
        torch.cuda.synchronize()
        prof = ctx.profile_read(); ctx.profile(False)
        lk = {l: ms / n for (s, l), (ms, n) in prof.items() if s == lkf.LKFLOW_STAGE_LK}
        print(f"P={P} th={th or 'auto':>5}: lk_L0 {lk[0]:.4f} ms  lk_L1 {lk.get(1, 0):.4f}  lk_L2 {lk.get(2, 0):.4f}", flush=True)
