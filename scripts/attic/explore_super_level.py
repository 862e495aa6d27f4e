"""Dev measurement: how many PAIRS OF 512-ROW SUPER TILES of the ordered image hold any pair whose 32-wide bound reaches a
threshold (what the int8 first pass could skip wholesale if it knew), and how many a centre + radius bound on the super
tiles (or on their four 128-row tiles) would flag.  usage: explore_super_level.py [n] [rank] [clusters]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
from arrowspace_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 8
clusters = int(sys.argv[3]) if len(sys.argv) > 3 else 256
seed = 42 if (rank, clusters) == (8, 256) else 43
d = 128
X, _ = synth.mixture_device(n, d, "cuda", seed=seed, n_clusters=clusters, rank=rank)
U = (X / X.norm(dim=1, keepdim=True)).float(); del X
g0 = torch.Generator(device="cuda"); g0.manual_seed(1)
w, V = torch.linalg.eigh((U.T @ U).double() / n)
V = V.flip(1).float()[:, :31]
Z = U @ V
ZT = torch.cat([Z, (1.0 - (Z * Z).sum(dim=1)).clamp_min(0).sqrt()[:, None]], dim=1)
def assign(points, piv):
    out = torch.empty(points.shape[0], dtype=torch.long, device="cuda")
    for a in range(0, points.shape[0], 65536): out[a:a + 65536] = (points[a:a + 65536] @ piv.T).argmax(dim=1)
    return out
piv = ZT[torch.randperm(n, device="cuda", generator=g0)[:1024]]
sub = ZT[torch.randperm(n, device="cuda", generator=g0)[:256]]
key = assign(ZT, piv) * 256 + assign(ZT, sub)
perm = torch.argsort(key, stable=True)
Up = U[perm].half(); Zp = ZT[perm]
S = Up[::32]
theta = torch.empty(n, device="cuda")
for a in range(0, n, 16384):
    C = (Up[a:a + 16384] @ S.T).float()
    ii = torch.arange(a, min(a + 16384, n), device="cuda")
    own = (ii % 32 == 0)
    C[own.nonzero()[:, 0], (ii[own] // 32)] = -2.0
    theta[a:a + 16384] = C.topk(4, dim=1).values[:, 3]
for G in (512, 128):
    ng = n // G
    Zg = Zp[:ng * G].view(ng, G, 32)
    tg = theta[:ng * G].view(ng, G).min(dim=1).values
    # centre + radius in the 31 coordinates, tails bounded by the largest tail of each group
    c = Zg[:, :, :31].mean(dim=1); r = (Zg[:, :, :31] - c[:, None, :]).norm(dim=2).max(dim=1).values; cn = c.norm(dim=1)
    tmax = Zg[:, :, 31].max(dim=1).values
    pick = torch.randperm(ng, device="cuda", generator=g0)[:48]
    true_any = 0; flagged = 0; total = 0
    for I in pick.tolist():
        zu = Zg[I]                                              # G x 32
        P = (zu @ Zp[:ng * G].T).view(G, ng, G)                 # bound of every pair of the row group with every row
        thr = torch.minimum(theta[I * G:(I + 1) * G][:, None, None], theta[:ng * G].view(ng, G)[None, :, :])
        true_any += int((P >= thr).any(dim=2).any(dim=0).sum())
        bound = c @ c[I] + r * cn[I] + r[I] * cn + r * r[I] + tmax * tmax[I]
        flagged += int((bound >= torch.minimum(tg, tg[I])).sum())
        total += ng
    print("groups of %d rows: %.4f of the group pairs hold a pair at its threshold; a centre + radius bound flags %.4f" % (G, true_any / total, flagged / total), flush=True)
    # one-sided angular bound: the group's rows sit in a cone of half angle a_I around its mean direction u_I (32 coordinates,
    # unit vectors), and the EXACT largest product of the other group's rows with u_I is known (an n x groups product, 1 / G of
    # the pass): cos(i, j) <= cos(max(0, angle(u_I, z_j) - a_I)); both orientations, the smaller bound counts
    Zu = Zg / Zg.norm(dim=2, keepdim=True).clamp_min(1e-30)
    u = Zu.mean(dim=1); u = u / u.norm(dim=1, keepdim=True).clamp_min(1e-30)
    cosa = (Zu * u[:, None, :]).sum(dim=2).clamp(-1, 1).min(dim=1).values        # cos of the cone's half angle
    aI = torch.acos(cosa)
    flagged2 = 0; total2 = 0
    for I in pick.tolist():
        # M1[J] = max_j z_j . u_I (rows of J against I's direction), M2[J] = max_i z_i . u_J (rows of I against J's direction)
        M1 = (Zu.view(ng * G, 32) @ u[I]).view(ng, G).max(dim=1).values.clamp(-1, 1)
        M2 = (Zu[I] @ u.T).max(dim=0).values.clamp(-1, 1)
        b1 = torch.cos((torch.acos(M1) - aI[I]).clamp_min(0))
        b2 = torch.cos((torch.acos(M2) - aI).clamp_min(0))
        bound = torch.minimum(b1, b2)
        # (the rows are not exactly unit in the 32 coordinates after quantisation; here they are: the bound is on z_i . z_j itself)
        flagged2 += int((bound >= torch.minimum(tg, tg[I])).sum()); total2 += ng
    print("    one-sided cone bound (exact far side): flags %.4f; cone half angle: median %.3f rad, 90 %% %.3f" % (
        flagged2 / total2, float(aI.median()), float(aI.quantile(0.9))), flush=True)
