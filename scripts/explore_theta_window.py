"""Dev measurement: how far the threshold of a row (4th best cosine against a 1-in-32 sample) moves when the sample is
restricted to rows near it in the join's image order (rows grouped by nearest pivot), or to the groups of the pivots most
similar to its own.  usage: explore_theta_window.py [n] [d]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
from arrowspace_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
X, _ = synth.mixture_device(n, d, "cuda", seed=42)
U = (X / X.norm(dim=1, keepdim=True)).float(); del X
g = torch.Generator(device="cuda"); g.manual_seed(1)
piv_ids = torch.randperm(n, device="cuda", generator=g)[:256]
piv = U[piv_ids]
assign = torch.empty(n, dtype=torch.long, device="cuda")
for a in range(0, n, 65536): assign[a:a + 65536] = (U[a:a + 65536] @ piv.T).argmax(dim=1)
perm = torch.argsort(assign, stable=True)
Up = U[perm]; gp = assign[perm]                      # image order, group of every position
samp = torch.arange(0, n, 32, device="cuda")          # sample = every 32nd position
S = Up[samp]; gs = gp[samp]
rows = torch.randperm(n, device="cuda", generator=g)[:4096]
C = Up[rows] @ S.T
C[(samp[None, :] == rows[:, None])] = -2.0
th_all = C.topk(4, dim=1).values[:, 3]
print("theta over the whole sample: mean %.4f" % th_all.mean())
pos_s = samp[None, :].expand(len(rows), -1)
for w in (2048, 4096, 8192, 16384, 65536):
    m = (pos_s - rows[:, None]).abs() <= w
    th = torch.where(m, C, torch.full_like(C, -2.0)).topk(4, dim=1).values[:, 3]
    diff = th_all - th
    print("window +-%6d positions: %5.0f sample rows; theta lower by > 1e-3 for %.3f of the rows, > 1e-2 for %.3f, > 5e-2 for %.4f, max %.3f" % (
        w, m.sum(dim=1).float().mean(), (diff > 1e-3).float().mean(), (diff > 1e-2).float().mean(), (diff > 5e-2).float().mean(), diff.max()))
PP = piv @ piv.T
for m_near in (1, 2, 4, 8, 16):
    near = PP.topk(m_near, dim=1).indices                 # groups of the most similar pivots (own one included)
    ok = torch.zeros(256, 256, dtype=torch.bool, device="cuda"); ok.scatter_(1, near, True)
    m = ok[gp[rows]][:, gs]
    th = torch.where(m, C, torch.full_like(C, -2.0)).topk(4, dim=1).values[:, 3]
    diff = th_all - th
    print("sample rows of the %2d nearest groups: %5.0f sample rows; theta lower by > 1e-3 for %.3f of the rows, > 1e-2 for %.3f, > 5e-2 for %.4f, max %.3f" % (
        m_near, m.sum(dim=1).float().mean(), (diff > 1e-3).float().mean(), (diff > 1e-2).float().mean(), (diff > 5e-2).float().mean(), diff.max()))
