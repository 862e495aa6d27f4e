"""Dev estimate: which share of the join's (64-row wave group) x (256-column stage) products a cone bound rules out when the
rows are ordered by (nearest of P sampled pivots, angle to it).  usage: explore_cone_pruning.py [n] [d] [pivots] [angle buckets]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth, distributed as D
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
P = int(sys.argv[3]) if len(sys.argv) > 3 else 256
NB = int(sys.argv[4]) if len(sys.argv) > 4 else 4
Xd, cent = synth.mixture_device(n, d, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
ctx.set_option("join_row_order", 0)
theta = D.HipOps(ctx).join_thresholds(Xd, 0, n, gp).double()
U = Xd / Xd.norm(dim=1, keepdim=True)
g = torch.Generator(device="cuda"); g.manual_seed(1)
Uf = U.float()
def flat(Pn):
    piv = Uf[torch.randperm(n, device="cuda", generator=g)[:Pn]]
    return (Uf @ piv.T).argmax(1)
def tree(P1, P2):
    a1 = flat(P1)
    out = torch.zeros(n, dtype=torch.long, device="cuda")
    for c in range(P1):
        idx = (a1 == c).nonzero().squeeze(1)
        if idx.numel() == 0: continue
        piv = Uf[idx[torch.randperm(idx.numel(), device="cuda", generator=g)[:P2]]]
        out[idx] = c * P2 + (Uf[idx] @ piv.T).argmax(1)
    return out
for name, key in (("flat %d" % P, flat(P)), ("tree 32x32", tree(32, 32)), ("tree 64x64", tree(64, 64)), ("tree 256x16", tree(256, 16))):
    perm = torch.argsort(key, stable=True)
    Up, th = U[perm], theta[perm]
    def tiles(rows):
        T = n // rows
        Ut = Up[:T * rows].view(T, rows, d)
        c = Ut.sum(1); c = c / c.norm(dim=1, keepdim=True)
        cosa = (Ut * c[:, None, :]).sum(2).min(1).values.clamp(-1, 1)
        return c, torch.acos(cosa), torch.acos(th[:T * rows].view(T, rows).min(1).values.clamp(-1, 1))
    for rI, rJ in ((64, 256), (512, 256)):
        cI, aI, tI = tiles(rI)
        cJ, aJ, tJ = tiles(rJ)
        sel = torch.arange(0, cI.shape[0], 13, device="cuda")
        phi = torch.acos((cI[sel] @ cJ.T).clamp(-1, 1))
        tau = torch.maximum(tI[sel][:, None], tJ[None, :])
        need = phi - aI[sel][:, None] - aJ[None, :] <= tau + 1e-3
        print("%-12s (%d x %d): must be contracted %.3f  (cone half-angles %.1f / %.1f deg, threshold angle %.1f deg)" % (
            name, rI, rJ, need.double().mean().item(), np.degrees(aI.mean().item()), np.degrees(aJ.mean().item()), np.degrees(tI.mean().item())), flush=True)
