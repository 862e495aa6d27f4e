"""Dev experiment: how the join's time depends on the quality of the row grouping (rows permuted on the host side, the
library's own reordering switched off).  usage: explore_row_order.py [n]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
X, cent = synth.mixture_device(n, 128, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
U = (X / X.norm(dim=1, keepdim=True)).float()
g = torch.Generator(device="cuda"); g.manual_seed(1)
def assign(P, lloyd):
    piv = U[torch.randperm(n, device="cuda", generator=g)[:P]]
    for _ in range(lloyd):
        a = (U @ piv.T).argmax(1)
        piv = torch.zeros(P, 128, device="cuda").index_add_(0, a, U)
        piv = piv / piv.norm(dim=1, keepdim=True).clamp_min(1e-30)
    return (U @ piv.T).argmax(1)
def tree(P1, P2):
    a1 = assign(P1, 0)
    out = torch.zeros(n, dtype=torch.long, device="cuda")
    for c in range(P1):
        idx = (a1 == c).nonzero().squeeze(1)
        if idx.numel() == 0: continue
        piv = U[idx[torch.randperm(idx.numel(), device="cuda", generator=g)[:P2]]]
        out[idx] = c * P2 + (U[idx] @ piv.T).argmax(1)
    return out
variants = {"as given": None, "true component": torch.arange(n, device="cuda") % 256, "256 pivots": assign(256, 0), "256 + 1 Lloyd": assign(256, 1),
            "1024 pivots": assign(1024, 0), "tree 16x16": tree(16, 16), "tree 32x32": tree(32, 32), "512 pivots": assign(512, 0)}
ctx.set_option("join_row_order", 0)
for name, a in variants.items():
    Xp = X if a is None else X[torch.argsort(a, stable=True)].contiguous()
    best = None
    for rep in range(3):
        ctx.profile(True)
        A.knn_cosine(Xp, gp, ctx=ctx); torch.cuda.synchronize()
        prof = ctx.profile_all()
        t = prof["knn_mfma_join_f16"][0]
        best = t if best is None else min(best, t)
    print("%-16s join %.1f ms  rerank %.1f" % (name, best, prof["knn_rerank"][0]), flush=True)
    del Xp
