"""Per-stage times of the headline step when the motion VARIES from pixel to pixel (rotation + zoom about the image centre, a few pixels at
the borders) instead of the synthetic workload's uniform translation: the fused kernel's W role reads its four bilinear weights from an LDS
table indexed by the two 5-bit fractions of the sample position -- one entry for the whole wave under a uniform translation (a broadcast
read), up to 64 different ones here.  usage: stage_times_varied_flow.py [pairs] [rows] [cols] [levels] [win] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "optical-flow_amd"))
import torch
import torch.nn.functional as F
import lkflow_amd as lk
import synth
P, H, W, L, win, reps = (int(x) for x in (sys.argv[1:7] + ["32", "1080", "1920", "3", "15", "12"][len(sys.argv) - 1:]))
prev, nxt_t = synth.frame_pairs_torch(P, H, W, "cuda", seed=1234)
ys, xs = torch.meshgrid(torch.linspace(-1, 1, H, dtype=torch.float64, device="cuda"), torch.linspace(-1, 1, W, dtype=torch.float64, device="cuda"), indexing="ij")
ang, zoom = 0.004, 1.003                                       # <= ~4.8 px of motion at the corners of a 1080p frame
gx = zoom * (xs * torch.cos(torch.tensor(ang)) - ys * (H / W) * torch.sin(torch.tensor(ang)))
gy = zoom * (xs * (W / H) * torch.sin(torch.tensor(ang)) + ys * torch.cos(torch.tensor(ang)))
grid = torch.stack([gx, gy], dim=-1)[None].expand(P, H, W, 2)
nxt_v = F.grid_sample(prev[:, None], grid, mode="bilinear", padding_mode="border", align_corners=True)[:, 0].contiguous()
ctx = lk.Context(0)
for name, nxt in (("uniform translation", nxt_t), ("rotation + zoom    ", nxt_v)):
    U = torch.empty_like(prev); V = torch.empty_like(prev)
    for _ in range(40): ctx.compute_batch(prev, nxt, L, win, out=(U, V))
    torch.cuda.synchronize()
    ctx.profile()
    for _ in range(reps): ctx.compute_batch(prev, nxt, L, win, out=(U, V))
    torch.cuda.synchronize()
    pr = ctx.profile_read(); ctx.profile(False)
    names = {0: "pyrdown", 1: "warp", 2: "lk"}
    tot = sum(ms for ms, n in pr.values()) / reps
    print(name, " ".join(f"{names[st]}{lv}={ms / reps:.3f}" for (st, lv), (ms, n) in sorted(pr.items())), f"| sum={tot:.3f} ms/step {P * H * W / tot / 1e3:.0f} Mpx/s",
          f"| median |flow| {torch.sqrt(U * U + V * V).median().item():.2f} px")
