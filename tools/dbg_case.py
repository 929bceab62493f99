"""Debug: one failing random case of tests/test_gpu_parity.py::test_random_small_shapes_against_oracle, on the paths
selected by the environment (LKFLOW_LIB + knobs) and by --fuse."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "optical-flow_amd"))
import torch
import lkflow_amd as lk
from oracle import oracle as O
O.build()
target = int(sys.argv[1]); fuse = int(sys.argv[2])
rng = np.random.default_rng(2026)
wins = [3, 3, 5, 7, 9, 15, 21, 27, 31]
for it in range(60):
    lv_target = int(rng.integers(1, 5))
    h = int(rng.integers(1, 40)) << (lv_target - 1)
    w = int(rng.integers(1, 40)) << (lv_target - 1)
    levels = min(lv_target, lk.getMaxLayer(w, h))
    win = wins[int(rng.integers(0, len(wins)))]
    a = rng.random((2, h, w)); b = rng.random((2, h, w))
    a[1, : h // 2] = 0.25; b[1, : h // 2] = 0.25
    if it == target: break
print("case", it, h, w, levels, win, "fuse", fuse, "lib", os.environ.get("LKFLOW_LIB"))
ctx = lk.Context(0)
ctx.set_option(lk.LKFLOW_OPT_FUSE_WARP, fuse)
for L in range(1, levels + 1):
    U, V = ctx.compute_batch(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), L, win)
    U, V = U.cpu().numpy(), V.cpu().numpy()
    for p in range(2):
        uo, vo = O.coarse_to_fine(a[p], b[p], L, win)
        scale = np.maximum(1.0, np.maximum(np.abs(uo), np.abs(vo)))
        e = np.maximum(np.abs(U[p] - uo), np.abs(V[p] - vo)) / scale
        bad = np.argwhere(e > 1e-6)
        print(f"L={L} p={p}: max rel err {e.max():.3e}  bad px {len(bad)}  max|uo| {np.abs(uo).max():.3e}  first bad {bad[:6].tolist()}")
np.save(os.path.join(ROOT, "gpurun_out", "r3a", f"U_fuse{fuse}_{'dbg' if os.environ.get('LKFLOW_LIB') else 'rel'}.npy"), np.stack([U, V]))
