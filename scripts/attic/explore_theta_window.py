"""Dev measurement: how far the threshold of a row (4th best cosine against a 1-in-32 sample) moves when the sample is
restricted to the rows of the pivot groups nearest to the row's own, for the order found on the rotated coordinates with
1024 pivots (rotate_wide.hip / row_order.hip).  usage: explore_theta_window.py [n] [d]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
from arrowspace_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
X, _ = synth.mixture_device(n, d, "cuda", seed=42)
U = (X / X.norm(dim=1, keepdim=True)).float(); del X
g = torch.Generator(device="cuda"); g.manual_seed(1)
w, V = torch.linalg.eigh((U.T @ U).double() / n)
V = V.flip(1).float()[:, :31]
Z = U @ V
ZT = torch.cat([Z, (1.0 - (Z * Z).sum(dim=1)).clamp_min(0).sqrt()[:, None]], dim=1)
for npiv in (256, 1024):
    piv = ZT[torch.randperm(n, device="cuda", generator=g)[:npiv]]
    assign = torch.empty(n, dtype=torch.long, device="cuda")
    for a in range(0, n, 65536): assign[a:a + 65536] = (ZT[a:a + 65536] @ piv.T).argmax(dim=1)
    perm = torch.argsort(assign, stable=True)
    Up = U[perm]; gp = assign[perm]
    samp = torch.arange(0, n, 32, device="cuda")
    S = Up[samp]; gs = gp[samp]
    rows = torch.randperm(n, device="cuda", generator=g)[:4096]
    C = Up[rows] @ S.T
    C[(samp[None, :] == rows[:, None])] = -2.0
    th_all = C.topk(4, dim=1).values[:, 3]
    PP = piv @ piv.T
    for m_near in (4, 8, 16, 32, 64):
        near = PP.topk(m_near, dim=1).indices
        ok = torch.zeros(npiv, npiv, dtype=torch.bool, device="cuda"); ok.scatter_(1, near, True)
        m = ok[gp[rows]][:, gs]
        th = torch.where(m, C, torch.full_like(C, -2.0)).topk(4, dim=1).values[:, 3]
        diff = th_all - th
        print("%4d pivots, sample rows of the %2d nearest groups: %5.0f sample rows (%.3f of the sample); theta lower by > 1e-3 for %.4f of the rows, > 1e-2 for %.4f, max %.3f" % (
            npiv, m_near, m.sum(dim=1).float().mean(), m.float().mean(), (diff > 1e-3).float().mean(), (diff > 1e-2).float().mean(), diff.max()))
