"""Dev measurement: would a first pass on GROUPS of adjacent image rows (centre + radius in the 32 rotated coordinates) in
front of the int8 first pass pay?  Level 1 contracts n / g centres (1 / g^2 of the products) and flags regions of 32 x 32
groups; level 2 (today's pass) visits the flagged regions only.  Printed: the share of regions level 1 flags, for g = 2, 4, 8,
against the share of 32 x 32 row blocks that hold a pair reaching a threshold (what today's pass flags).
usage: explore_two_level.py [n] [rank] [clusters]"""
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
ZT = torch.cat([Z, (1.0 - (Z * Z).sum(dim=1)).clamp_min(0).sqrt()[:, None]], dim=1)   # 32 coordinates, norm 1
print("tail energy %.3f" % float((ZT[:, 31] ** 2).mean()))
def assign(points, piv):
    out = torch.empty(points.shape[0], dtype=torch.long, device="cuda")
    for a in range(0, points.shape[0], 65536): out[a:a + 65536] = (points[a:a + 65536] @ piv.T).argmax(dim=1)
    return out
piv = ZT[torch.randperm(n, device="cuda", generator=g0)[:1024]]
sub = ZT[torch.randperm(n, device="cuda", generator=g0)[:256]]
key = assign(ZT, piv) * 256 + assign(ZT, sub)
perm = torch.argsort(key, stable=True)
Up = U[perm].half(); Zp = ZT[perm]
# thresholds: 4th best cosine against a 1-in-32 sample
S = Up[::32]
theta = torch.empty(n, device="cuda")
for a in range(0, n, 16384):
    C = (Up[a:a + 16384] @ S.T).float()
    ii = torch.arange(a, min(a + 16384, n), device="cuda")
    own = (ii % 32 == 0)
    C[own.nonzero()[:, 0], (ii[own] // 32)] = -2.0
    theta[a:a + 16384] = C.topk(4, dim=1).values[:, 3]
print("thresholds: mean %.3f, 1 %% quantile %.3f" % (float(theta.mean()), float(theta.quantile(0.01))))
rows_blocks = torch.randperm(n // 256, device="cuda", generator=g0)[:64]
# today: 32 x 32 row blocks with a pair whose bound reaches min(theta_u, theta_v)
nb = n // 32
flag_now = 0; total_now = 0
tb = theta[:nb * 32].view(nb, 32)
for rb in rows_blocks.tolist():
    a = rb * 256
    for q in range(8):
        zu = Zp[a + 32 * q:a + 32 * q + 32]; tu = theta[a + 32 * q:a + 32 * q + 32]
        P = (zu @ Zp[:nb * 32].T).view(32, nb, 32)                   # bound of every pair
        thr = torch.minimum(tu[:, None, None], tb[None, :, :])
        flag_now += int(((P >= thr).any(dim=2).any(dim=0)).sum()); total_now += nb
print("today: %.4f of the 32 x 32 blocks hold a pair at its threshold" % (flag_now / total_now))
for g in (2, 4, 8):
    ng = n // g
    Zg = Zp[:ng * g].view(ng, g, 32)
    c = Zg.mean(dim=1); r = (Zg - c[:, None, :]).norm(dim=2).max(dim=1).values; cn = c.norm(dim=1)
    tg = theta[:ng * g].view(ng, g).min(dim=1).values
    R = 32 * g                                                       # rows of a region side
    nreg = n // R
    flagged = 0; total = 0; gp_pass = 0; gp_total = 0
    for rb in rows_blocks.tolist():
        a = (rb * 256) // R * R
        ga = a // g
        cu = c[ga:ga + 32]; ru = r[ga:ga + 32]; nu = cn[ga:ga + 32]; tu = tg[ga:ga + 32]
        B = cu @ c.T + ru[:, None] * cn[None, :] + nu[:, None] * r[None, :] + ru[:, None] * r[None, :]
        ok = B >= torch.minimum(tu[:, None], tg[None, :])
        gp_pass += int(ok.sum()); gp_total += ok.numel()
        okr = ok[:, :nreg * 32].view(32, nreg, 32).any(dim=2).any(dim=0)
        flagged += int(okr.sum()); total += nreg
    # the same at the granularity of the first pass's stages: 512 rows x 512 columns
    f512 = 0; t512 = 0
    n512 = n // 512; G = 512 // g
    for rb in rows_blocks.tolist():
        a = (rb * 256) // 512 * 512
        ga = a // g
        cu = c[ga:ga + G]; ru = r[ga:ga + G]; nu = cn[ga:ga + G]; tu = tg[ga:ga + G]
        B = cu @ c.T + ru[:, None] * cn[None, :] + nu[:, None] * r[None, :] + ru[:, None] * r[None, :]
        ok = B >= torch.minimum(tu[:, None], tg[None, :])
        f512 += int(ok[:, :n512 * G].view(G, n512, G).any(dim=2).any(dim=0).sum()); t512 += n512
    print("groups of %d rows: %.4f of the 512 x 512 stages flagged" % (g, f512 / t512))
    print("groups of %d rows: radius mean %.3f; %.4f of the group pairs pass, %.4f of the %d x %d regions flagged -> level 1 %.3f + level 2 %.3f = %.3f of today's products" % (
        g, float(r.mean()), gp_pass / gp_total, flagged / total, R, R, 1 / g ** 2, flagged / total, 1 / g ** 2 + flagged / total), flush=True)
