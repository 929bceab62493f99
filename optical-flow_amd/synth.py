"""Synthetic "translated noise" frames: the workload BASELINE.json's configs name.

base  = uniform [0,1) noise smoothed by a 9x9 sigma=1.5 Gaussian (mirrors the reference's own
        pre-blur, ParallelVideoPyr.cpp:790), renormalised to [0,1]
next  = base translated by (dx, dy) = (+1.5, -0.75) px with bilinear interpolation (edge clamp)

numpy version: small deterministic inputs for the parity tests (uploaded to the GPU).
torch version: same recipe generated directly in HBM for bench.py (full-size batches).
"""
import numpy as np


def _gauss_kernel(ksize=9, sigma=1.5):
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2
    k = np.exp(-(x * x) / (2 * sigma * sigma))
    return k / k.sum()


def smooth_noise(h, w, seed=1234):
    rng = np.random.Generator(np.random.PCG64(seed))
    a = rng.random((h, w))
    k = _gauss_kernel()
    r = len(k) // 2
    p = np.pad(a, ((0, 0), (r, r)), mode="reflect")
    a = sum(k[i] * p[:, i:i + w] for i in range(len(k)))
    p = np.pad(a, ((r, r), (0, 0)), mode="reflect")
    a = sum(k[i] * p[i:i + h, :] for i in range(len(k)))
    a -= a.min()
    a /= a.max()
    return a


def translate(img, dx, dy):
    """out(y,x) = img(y - dy, x - dx), bilinear, clamped at the edges."""
    h, w = img.shape
    ys = np.clip(np.arange(h, dtype=np.float64) - dy, 0, h - 1)
    xs = np.clip(np.arange(w, dtype=np.float64) - dx, 0, w - 1)
    y0 = np.floor(ys).astype(np.int64); x0 = np.floor(xs).astype(np.int64)
    y1 = np.minimum(y0 + 1, h - 1); x1 = np.minimum(x0 + 1, w - 1)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    return ((1 - fy) * ((1 - fx) * img[y0][:, x0] + fx * img[y0][:, x1]) +
            fy * ((1 - fx) * img[y1][:, x0] + fx * img[y1][:, x1]))


def frame_pair(h, w, seed=1234, shift=(1.5, -0.75)):
    a = smooth_noise(h, w, seed)
    return a, translate(a, shift[0], shift[1])


def video(n_frames, h, w, seed=1234, shift=(1.5, -0.75)):
    """Cumulative translation per frame (frame k = base shifted by k*shift)."""
    a = smooth_noise(h, w, seed)
    return np.stack([translate(a, k * shift[0], k * shift[1]) if k else a for k in range(n_frames)])


# ---- torch (device) version for bench.py ---------------------------------------------------------
def frame_pairs_torch(n_pairs, h, w, device, seed=1234, shift=(1.5, -0.75)):
    """[P,H,W] prev, next float64 tensors generated on `device` (each pair has its own noise)."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    k = torch.tensor(_gauss_kernel(), dtype=torch.float64, device=device)
    r = k.numel() // 2
    prev = torch.empty((n_pairs, h, w), dtype=torch.float64, device=device)
    nxt = torch.empty_like(prev)
    ys = (torch.arange(h, dtype=torch.float64, device=device) - shift[1]).clamp(0, h - 1)
    xs = (torch.arange(w, dtype=torch.float64, device=device) - shift[0]).clamp(0, w - 1)
    y0 = ys.floor().long(); x0 = xs.floor().long()
    y1 = (y0 + 1).clamp(max=h - 1); x1 = (x0 + 1).clamp(max=w - 1)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    for p in range(n_pairs):
        a = torch.rand((1, 1, h, w), dtype=torch.float64, device=device, generator=g)
        a = F.conv2d(F.pad(a, (r, r, 0, 0), mode="reflect"), k.view(1, 1, 1, -1))
        a = F.conv2d(F.pad(a, (0, 0, r, r), mode="reflect"), k.view(1, 1, -1, 1))
        a = a[0, 0]
        a = a - a.min()
        a = a / a.max()
        prev[p] = a
        nxt[p] = ((1 - fy) * ((1 - fx) * a[y0][:, x0] + fx * a[y0][:, x1]) +
                  fy * ((1 - fx) * a[y1][:, x0] + fx * a[y1][:, x1]))
    return prev, nxt


def video_torch(n_frames, h, w, device, seed=1234, shift=(1.5, -0.75)):
    """[F,H,W] float64 frames generated on `device`: one smoothed-noise base, frame k = frame k-1 translated by
    `shift` (bilinear, edge-clamped) -- the stream form of the synthetic workload ("video batches:
    cumulative translation per frame").  Consecutive frames form the F-1 pairs."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    k = torch.tensor(_gauss_kernel(), dtype=torch.float64, device=device)
    r = k.numel() // 2
    ys = (torch.arange(h, dtype=torch.float64, device=device) - shift[1]).clamp(0, h - 1)
    xs = (torch.arange(w, dtype=torch.float64, device=device) - shift[0]).clamp(0, w - 1)
    y0 = ys.floor().long(); x0 = xs.floor().long()
    y1 = (y0 + 1).clamp(max=h - 1); x1 = (x0 + 1).clamp(max=w - 1)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    a = torch.rand((1, 1, h, w), dtype=torch.float64, device=device, generator=g)
    a = F.conv2d(F.pad(a, (r, r, 0, 0), mode="reflect"), k.view(1, 1, 1, -1))
    a = F.conv2d(F.pad(a, (0, 0, r, r), mode="reflect"), k.view(1, 1, -1, 1))
    a = a[0, 0]
    a = a - a.min()
    a = a / a.max()
    frames = torch.empty((n_frames, h, w), dtype=torch.float64, device=device)
    frames[0] = a
    for f in range(1, n_frames):
        a = ((1 - fy) * ((1 - fx) * a[y0][:, x0] + fx * a[y0][:, x1]) +
             fy * ((1 - fx) * a[y1][:, x0] + fx * a[y1][:, x1]))
        frames[f] = a
    return frames

