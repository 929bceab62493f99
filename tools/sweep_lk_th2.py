"""Alternating repeat of a few LK strip heights (LKFLOW_OPT_TILE_ROWS) in one process: is a taller strip than the
whole-rounds model picks faster at level 0?   python tools/sweep_lk_th2.py [pairs] [config] [th ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "optical-flow_amd"))
import torch
import lkflow_amd as lkf
import synth
import bench

P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
name = sys.argv[2] if len(sys.argv) > 2 else "1080p"
ths = [int(t) for t in sys.argv[3:]] or [0, 136, 180, 216, 270]
cols, rows, levels, win, _ = bench.CONFIGS[name]
dev = torch.device("cuda", 0)
prev, nxt = synth.frame_pairs_torch(P, rows, cols, dev, seed=1)
U, V = torch.empty_like(prev), torch.empty_like(prev)
with lkf.Context(0) as ctx:
    ctx.reserve(rows, cols, levels, P)
    stream = torch.cuda.current_stream()
    for _ in range(20):
        ctx.compute_batch(prev, nxt, levels, win, out=(U, V), stream=stream)
    acc = {t: [] for t in ths}
    for rnd in range(5):
        for th in ths:
            ctx.set_option(lkf.LKFLOW_OPT_TILE_ROWS, th)
            for _ in range(2):
                ctx.compute_batch(prev, nxt, levels, win, out=(U, V), stream=stream)
            torch.cuda.synchronize()
            ctx.profile()
            for _ in range(10):
                ctx.compute_batch(prev, nxt, levels, win, out=(U, V), stream=stream)
            torch.cuda.synchronize()
            prof = ctx.profile_read(); ctx.profile(False)
            lk = {l: ms / n for (s, l), (ms, n) in prof.items() if s == lkf.LKFLOW_STAGE_LK}
            acc[th].append((lk[0], sum(lk.values())))
    for th in ths:
        l0 = sorted(a for a, _ in acc[th]); tot = sorted(b for _, b in acc[th])
        print(f"P={P} {name} th={th or 'auto':>5}: lk_L0 median {l0[2]:.4f} (min {l0[0]:.4f} max {l0[-1]:.4f})  all LK levels median {tot[2]:.4f}", flush=True)
