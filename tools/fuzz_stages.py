#!/usr/bin/env python3
"""One-off fuzz of the bit-exact stages (pyrDown, warp, blur+gray, frame difference, edge map, centroid) on random shapes."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "optical-flow_amd")); sys.path.insert(0, R)
import torch
import lkflow_amd as L
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = L.Context(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for it in range(n):
    h, w = int(rng.integers(1, 150)), int(rng.integers(1, 300))
    a = rng.standard_normal((2, h, w))
    got = ctx.pyrdown(dev(a)).cpu().numpy()
    for k in range(2):
        assert np.array_equal(got[k], O.pyr_down(a[k])), ("pyrdown", h, w)
    # strided source (row padding, odd base offset)
    pad = torch.zeros((2, h, w + 3), dtype=torch.float64, device="cuda"); pad[:, :, 1:w + 1] = dev(a)
    got = ctx.pyrdown(pad[:, :, 1:w + 1]).cpu().numpy()
    for k in range(2):
        assert np.array_equal(got[k], O.pyr_down(a[k])), ("pyrdown strided", h, w)
    he, we = 2 * int(rng.integers(1, 60)), 2 * int(rng.integers(1, 120))
    i1, i2 = rng.random((2, he, we)), rng.random((2, he, we))
    sc = float(rng.choice([0.5, 3.0, 40.0, 1e4]))
    cu, cv = rng.standard_normal((2, he // 2, we // 2)) * sc, rng.standard_normal((2, he // 2, we // 2)) * sc
    o1, o2 = ctx.warp(dev(i1), dev(i2), dev(cu), dev(cv))
    o1, o2 = o1.cpu().numpy(), o2.cpu().numpy()
    for k in range(2):
        upu, upv = O.pyr_up(cu[k]), O.pyr_up(cv[k])
        assert np.array_equal(o1[k], O.remap_linear(i1[k], O.build_map(upu, upv, 1.0))), ("warp1", he, we, sc)
        assert np.array_equal(o2[k], O.remap_linear(i2[k], O.build_map(upu, upv, -1.0))), ("warp2", he, we, sc)
    hb, wb = int(rng.integers(1, 90)), int(rng.integers(1, 150))
    ks = int(rng.choice([1, 3, 5, 7, 9, 11, 13, 15])); sg = float(rng.uniform(0.3, 4.0))
    f = rng.integers(0, 256, (3, hb, wb, 3)).astype(np.uint8)
    g = ctx.blur_gray(dev(f), ks, sg).cpu().numpy()
    for k in range(3):
        assert np.array_equal(g[k], O.blur_gray_bgr8(f[k], ks, sg)), ("blur", hb, wb, ks, sg)
    rate = float(rng.uniform(0.0, 1.2)); th = int(rng.integers(0, 60))
    e = ctx.diff_edge_map(dev(g), rate, th).cpu().numpy()
    for k in range(2):
        d = O.frame_diff_u8(g[k + 1], g[k], rate)
        assert np.array_equal(e[k], O.edge_map_u8(d, th)), ("edge", hb, wb, rate, th)
    t = float(rng.uniform(0.1, 30.0))
    c = ctx.flow_centroid(dev(cu), dev(cv), t).cpu().numpy()
    for k in range(2):
        cc, sx, sy = O.flow_centroid(cu[k], cv[k], t)
        assert tuple(c[k]) == (float(cc), sx, sy), ("centroid", t)
print(f"stage fuzz ok: {n} rounds")
