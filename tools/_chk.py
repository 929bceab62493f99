import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "optical-flow_amd"))
import numpy as np, torch
import lkflow_amd as lk, synth
from oracle import oracle as O
O.build()
ctx = lk.Context(0)
for win in [int(w) for w in sys.argv[1:]]:
    for (h, w) in ((96, 288), (200, 264)):
        a, b = synth.frame_pair(h, w, seed=win)
        a2 = a.copy(); b2 = b.copy(); a2[: h // 2] = 0.25; b2[: h // 2] = 0.25
        for (x, y) in ((a, b), (a2, b2)):
            U, V = ctx.compute_batch(torch.from_numpy(np.ascontiguousarray(x)).cuda()[None], torch.from_numpy(np.ascontiguousarray(y)).cuda()[None], 3, win)
            uo, vo = O.coarse_to_fine(x, y, 3, win)
            e = max(np.abs(U[0].cpu().numpy() - uo).max(), np.abs(V[0].cpu().numpy() - vo).max())
            assert e <= 1e-6, (win, h, w, e)
    print("win", win, "ok", e)
