import os, sys, numpy as np
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/optical-flow_amd")
import torch, lkflow_amd as lk
from oracle import oracle as O
O.build()
Y=np.load(ROOT+"/tests/golden/pacman_240x160_y.npz")["Y"]
ctx=lk.Context(0)
d=ctx.u8_to_f64(torch.from_numpy(Y).cuda())
U,V=ctx.compute_video(d,3,15)
torch.cuda.synchronize()
print("flow range", U.min().item(), U.max().item(), V.min().item(), V.max().item(), "finite", torch.isfinite(U).all().item())
f=d.cpu().numpy()
for p in (0,1,60,121,243):
    uo,vo=O.coarse_to_fine(f[p],f[p+1],3,15)
    du=np.abs(U[p].cpu().numpy()-uo); dv=np.abs(V[p].cpu().numpy()-vo)
    print(p, "max|dU| %.3e max|dV| %.3e"%(du.max(),dv.max()), "px > 1e-6:", int(((du>1e-6)|(dv>1e-6)).sum()), "of", du.size, " |U|max oracle %.3f"%np.abs(uo).max(), "nonzero flow px", int((uo!=0).sum()))
