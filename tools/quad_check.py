import os, sys
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(R, "optical-flow_amd")); sys.path.insert(0, R)
import lkflow_amd as L, synth
from oracle import oracle as O
with L.Context(0) as ctx:
    for (rows, cols, lv, win) in ((272, 480, 3, 31), (160, 240, 2, 31), (544, 960, 4, 31)):
        a, b = synth.frame_pair(rows, cols, seed=rows + win)
        u, v = ctx.computeFlow(a, b, lv, win)
        uo, vo = O.coarse_to_fine(a, b, lv, win)
        sc = max(1.0, np.abs(uo).max(), np.abs(vo).max())
        print(rows, cols, lv, win, "max|d| =", max(np.abs(u - uo).max(), np.abs(v - vo).max()), "scale", sc, flush=True)
