"""Fine landscape of the LK strip height per pyramid level (LKFLOW_OPT_TILE_ROWS forces one height at every level; each
level's launch is timed by its own HIP events): python tools/sweep_lk_th3.py [pairs] [config] [lo hi step]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "optical-flow_amd"))
import torch
import lkflow_amd as lkf
import synth
import bench

P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
name = sys.argv[2] if len(sys.argv) > 2 else "1080p"
lo, hi, st = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (24, 560, 6)
cols, rows, levels, win, _ = bench.CONFIGS[name]
ths = [0] + list(range(lo, hi + 1, st))
dev = torch.device("cuda", 0)
prev, nxt = synth.frame_pairs_torch(P, rows, cols, dev, seed=1)
U, V = torch.empty_like(prev), torch.empty_like(prev)
with lkf.Context(0) as ctx:
    ctx.reserve(rows, cols, levels, P)
    stream = torch.cuda.current_stream()
    for _ in range(20):
        ctx.compute_batch(prev, nxt, levels, win, out=(U, V), stream=stream)
    acc = {t: [] for t in ths}
    for rnd in range(3):
        for th in ths:
            ctx.set_option(lkf.LKFLOW_OPT_TILE_ROWS, th)
            ctx.compute_batch(prev, nxt, levels, win, out=(U, V), stream=stream)
            torch.cuda.synchronize()
            ctx.profile()
            for _ in range(6):
                ctx.compute_batch(prev, nxt, levels, win, out=(U, V), stream=stream)
            torch.cuda.synchronize()
            prof = ctx.profile_read(); ctx.profile(False)
            acc[th].append({l: ms / n for (s, l), (ms, n) in prof.items() if s == lkf.LKFLOW_STAGE_LK})
    for th in ths:
        med = {l: sorted(a[l] for a in acc[th])[1] for l in acc[th][0]}
        print(f"P={P} {name} th={th or 'auto':>5}: " + "  ".join(f"L{l} {med[l]:.4f}" for l in sorted(med)), flush=True)
