"""Dev measurement: what an 8-bit first pass of the symmetric join would let through on the bench data.
For int8 / fp8 images of the unit rows: the row error norm E (the bound is 2E + E^2), candidates per row against
theta - delta, rows whose K-th neighbour clears theta + delta, and the share of 64 x 32 blocks (rows in pivot order) that
hold a pair above min(theta_i, theta_j) - delta.  usage: explore_int8_prefilter.py [n] [d]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
from arrowspace_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
K = 17
X, _ = synth.mixture_device(n, d, "cuda", seed=42)
U = (X / X.norm(dim=1, keepdim=True)).float()
del X
g = torch.Generator(device="cuda"); g.manual_seed(1)
# thresholds: 4th best against a 1-in-32 sample
S = U[torch.randperm(n, device="cuda", generator=g)[: n // 32]]
theta = torch.empty(n, device="cuda")
for a in range(0, n, 16384):
    theta[a:a + 16384] = (U[a:a + 16384] @ S.T).topk(5, dim=1).values[:, 4]  # (self may be in the sample)
print("theta: mean %.4f min %.4f max %.4f" % (theta.mean(), theta.min(), theta.max()))
# images
absmax = U.abs().max()
step = absmax / 127.0
Ui8 = torch.round(U / step).clamp(-127, 127) * step
E_i8 = (Ui8 - U).norm(dim=1)
rowmax = U.abs().amax(dim=1, keepdim=True)
Ui8r = torch.round(U / (rowmax / 127.0)).clamp(-127, 127) * (rowmax / 127.0)
E_i8r = (Ui8r - U).norm(dim=1)
U8 = (U * 16.0).to(torch.float8_e4m3fn).float() / 16.0
E_f8 = (U8 - U).norm(dim=1)
Uh = U.half().float()
E_h = (Uh - U).norm(dim=1)
print("abs max %.4f  row max mean %.4f" % (absmax, rowmax.mean()))
for name, E in (("f16", E_h), ("int8 global", E_i8), ("int8 per row", E_i8r), ("e4m3 x16", E_f8)):
    print("%-14s E max %.5f mean %.5f -> delta %.5f" % (name, E.max(), E.mean(), 2 * E.max() + E.max() ** 2))
# candidates per row and certification against delta, on sampled rows
ids = torch.randperm(n, device="cuda", generator=g)[:2048]
C = U[ids] @ U.T
C[torch.arange(len(ids)), ids] = -2.0
top = C.topk(K, dim=1).values
tr = theta[ids]
for delta in (6e-4, 0.005, 0.01, 0.02, 0.03, 0.05, 0.075):
    either = (C >= (torch.minimum(tr[:, None], theta[None, :]) - delta)).sum(dim=1).float()
    own = (C >= (tr[:, None] - delta)).sum(dim=1).float()
    cert = (top[:, K - 1] > tr + delta).float().mean()
    print("delta %.4f: own list %.0f mean %.0f max; listed either side %.0f mean; certified %.4f" % (
        delta, own.mean(), own.max(), either.mean(), cert))
del C
# blocks in pivot order
piv = U[torch.randperm(n, device="cuda", generator=g)[:256]]
assign = torch.empty(n, dtype=torch.long, device="cuda")
for a in range(0, n, 65536):
    assign[a:a + 65536] = (U[a:a + 65536] @ piv.T).argmax(dim=1)
perm = torch.argsort(assign, stable=True)
Up = U[perm]; tp = theta[perm]
nb = n // 32
groups = torch.randint(0, n // 64, (128,), generator=torch.Generator().manual_seed(3))
for delta in (6e-4, 0.01, 0.02, 0.03, 0.05, 0.075):
    tot = 0.0
    tot256 = 0.0
    for gi in groups.tolist():
        r0 = gi * 64
        Cb = Up[r0:r0 + 64] @ Up[: nb * 32].T
        hit = Cb >= (torch.minimum(tp[r0:r0 + 64, None], tp[None, : nb * 32]) - delta)
        blk = hit.view(64, nb, 32).any(dim=2).any(dim=0)
        tot += blk.float().mean().item()
        tot256 += blk[: (nb // 8) * 8].view(-1, 8).any(dim=1).float().mean().item()
    print("delta %.4f: 64x32 blocks holding a pair %.4f; 64x256 stages holding one %.4f" % (delta, tot / len(groups), tot256 / len(groups)))

# ---- a rotated, truncated first pass: cos <= z_i . z_j + t_i t_j with z = the leading m coordinates in an orthonormal basis
# of the dominant subspace and t = the norm of what is left (rigorous for any orthonormal basis) ----
Cov = (U.T @ U).double() / n
w, V = torch.linalg.eigh(Cov)
V = V.flip(1).float(); w = w.flip(0)
print("leading eigenvalues (share of energy):", [round(float(x), 4) for x in (w[:8] / w.sum())], "cumulative 31: %.4f 63: %.4f" % (w[:31].sum() / w.sum(), w[:63].sum() / w.sum()))
for m in (15, 31, 63):
    Z = Up @ V[:, :m]
    t = (1.0 - (Z * Z).sum(dim=1)).clamp_min(0).sqrt()
    print("m=%d: tail norm mean %.4f max %.4f" % (m, t.mean(), t.max()))
    zmax = Z.abs().max(); stepz = zmax / 127.0
    Zq = torch.round(Z / stepz).clamp(-127, 127) * stepz
    Ez = (Zq - Z).norm(dim=1).max()
    tq = torch.ceil(t / stepz).clamp(max=127) * stepz
    dz = 2 * Ez + Ez * Ez
    print("   step %.5f Ez %.5f delta_z %.5f; t saturates: %d rows" % (stepz, Ez, dz, int((t / stepz > 127).sum())))
    Zq = torch.cat([Zq, tq[:, None]], dim=1)
    tot = 0.0
    for gi in groups.tolist():
        r0 = gi * 64
        Sb = Zq[r0:r0 + 64] @ Zq[: nb * 32].T
        hit = Sb >= (torch.minimum(tp[r0:r0 + 64, None], tp[None, : nb * 32]) - dz)
        tot += hit.view(64, nb, 32).any(dim=2).any(dim=0).float().mean().item()
    print("   64x32 blocks flagged %.4f" % (tot / len(groups)))

# ---- how the number of pivots of the row order (found on the 32 rotated coordinates) moves the share of flagged blocks ----
m = 31
Zall = U @ V[:, :m]
tall = (1.0 - (Zall * Zall).sum(dim=1)).clamp_min(0).sqrt()
ZT = torch.cat([Zall, tall[:, None]], dim=1)
for npiv in (256, 512, 1024, 2048):
    pid = torch.randperm(n, device="cuda", generator=g)[:npiv]
    pz = ZT[pid]
    asg = torch.empty(n, dtype=torch.long, device="cuda")
    for a in range(0, n, 65536): asg[a:a + 65536] = (ZT[a:a + 65536] @ pz.T).argmax(dim=1)
    pm = torch.argsort(asg, stable=True)
    Zp = ZT[pm]; tpp = theta[pm]
    stepz = 1.0 / 127.0
    Zq = torch.cat([torch.round(Zp[:, :m] / stepz).clamp(-127, 127), torch.ceil(Zp[:, m:] / stepz).clamp(max=127)], dim=1) * stepz
    Ez = (Zq[:, :m] - Zp[:, :m]).norm(dim=1).max(); dz = 2 * Ez + Ez * Ez
    tot = 0.0
    for gi in groups.tolist():
        r0 = gi * 64
        Sb = Zq[r0:r0 + 64] @ Zq[: nb * 32].T
        hit = Sb >= (torch.minimum(tpp[r0:r0 + 64, None], tpp[None, : nb * 32]) - dz)
        tot += hit.view(64, nb, 32).any(dim=2).any(dim=0).float().mean().item()
    print("pivots %4d: 64x32 blocks flagged %.4f" % (npiv, tot / len(groups)))
