"""Dev estimate: which share of the join's super-tile pairs a cone bound rules out when the rows are ordered by a cheap
clustering (nearest of P sampled pivots by cosine).  usage: explore_cone_pruning.py [n] [d] [pivots]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth, distributed as D
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
P = int(sys.argv[3]) if len(sys.argv) > 3 else 256
Xd, cent = synth.mixture_device(n, d, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
theta = D.HipOps(ctx).join_thresholds(Xd, 0, n, gp).double()
U = Xd / Xd.norm(dim=1, keepdim=True)
g = torch.Generator(device="cuda"); g.manual_seed(1)
for order in ("pivots", "pivots2"):
    piv = U[torch.randperm(n, device="cuda", generator=g)[:P]]
    if order == "pivots2":  # one Lloyd step on the sphere: pivots = cluster means
        a = (U.float() @ piv.float().T).argmax(1)
        piv = torch.zeros(P, d, dtype=torch.float64, device="cuda").index_add_(0, a, U)
        piv = piv / piv.norm(dim=1, keepdim=True).clamp_min(1e-30)
    assign = (U.float() @ piv.float().T).argmax(1)
    perm = torch.argsort(assign, stable=True)
    for tile in (512, 256):
        T = (n + tile - 1) // tile
        Up = U[perm]
        th = theta[perm]
        pad = T * tile - n
        if pad:
            Up = torch.cat([Up, Up[-1:].expand(pad, d)]); th = torch.cat([th, th[-1:].expand(pad)])
        Ut = Up.view(T, tile, d)
        c = Ut.sum(1); c = c / c.norm(dim=1, keepdim=True)
        cosa = (Ut * c[:, None, :]).sum(2).min(1).values.clamp(-1, 1)
        alpha = torch.acos(cosa)
        thmin = th.view(T, tile).min(1).values
        phi = torch.acos((c @ c.T).clamp(-1, 1))
        ub = torch.cos((phi - alpha[:, None] - alpha[None, :]).clamp_min(0))
        need = ub + 7e-4 >= torch.minimum(thmin[:, None], thmin[None, :])
        print("%s P=%d tile %d: pairs that must be contracted: %.3f of all (mean cone half-angle %.1f deg, theta min per tile mean %.3f)" % (
            order, P, tile, need.double().mean().item(), np.degrees(alpha.mean().item()), thmin.mean().item()), flush=True)
