"""Dev measurement: grouped thresholds on the hard mixture (rank 64 / 4 096 clusters), where the groups nearest by pivot
similarity miss a third of the rows' sample partners.  Which 16 groups should a group's sample hold?  Compared: nearest by
pivot similarity (what join_thresholds_f16_grouped does) against the groups its sampled rows' best sample partners live in
(votes from a sample x sample product, 1/32 of the plain pass).  usage: explore_theta_votes.py [n] [rank] [clusters]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
from arrowspace_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 64
clusters = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
d = 128
X, _ = synth.mixture_device(n, d, "cuda", seed=43, n_clusters=clusters, rank=rank)
U = (X / X.norm(dim=1, keepdim=True)).float(); del X
g = torch.Generator(device="cuda"); g.manual_seed(1)
w, V = torch.linalg.eigh((U.T @ U).double() / n)
V = V.flip(1).float()[:, :31]
Z = U @ V
ZT = torch.cat([Z, (1.0 - (Z * Z).sum(dim=1)).clamp_min(0).sqrt()[:, None]], dim=1)
npiv = 1024
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
def report(name, near):
    ok = torch.zeros(npiv, npiv, dtype=torch.bool, device="cuda"); ok.scatter_(1, near, True)
    m = ok[gp[rows]][:, gs]
    th = torch.where(m, C, torch.full_like(C, -2.0)).topk(4, dim=1).values[:, 3]
    diff = th_all - th
    print("%-44s %5.0f sample rows; theta lower by > 5e-3 for %.4f of the rows, > 1e-2 for %.4f, max %.3f" % (
        name, m.sum(dim=1).float().mean(), (diff > 5e-3).float().mean(), (diff > 1e-2).float().mean(), diff.max()), flush=True)
PP = piv @ piv.T
for m_near in (16, 32, 64): report("nearest %d by pivot similarity" % m_near, PP.topk(m_near, dim=1).indices)
# votes: the groups of the 4 / 8 best sample partners of every sampled row
ns = S.shape[0]
for kbest in (4, 8):
    votes = torch.zeros(npiv, npiv, device="cuda")
    for a in range(0, ns, 4096):
        Cs = S[a:a + 4096] @ S.T
        Cs[torch.arange(Cs.shape[0], device="cuda"), torch.arange(a, a + Cs.shape[0], device="cuda")] = -2.0
        idx = Cs.topk(kbest, dim=1).indices
        votes.index_put_((gs[a:a + 4096][:, None].expand(-1, kbest).reshape(-1), gs[idx].reshape(-1)), torch.ones(idx.numel(), device="cuda"), accumulate=True)
    votes += torch.eye(npiv, device="cuda") * 1e6  # (its own group first)
    for m_near in (16, 32, 64): report("top %d groups by votes of the best %d partners" % (m_near, kbest), votes.topk(m_near, dim=1).indices)
    sym = votes + votes.T
    for m_near in (16, 32): report("top %d by symmetric votes (best %d)" % (m_near, kbest), sym.topk(m_near, dim=1).indices)
