"""What the lambda leg of a bench step (ShardedArrowSpace.build_lambdas) spends where: the F x F graph of the centroids, the per-item
lambdas, the logged statistics, the index.  usage: python scripts/attic/lambda_leg_parts.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth, distributed as D
n, d = 1_000_000, 128
ctx = A.default_context()
Xd, cent = synth.mixture_device(n, d, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
ops = D.HipOps(ctx)
sas = D.ShardedArrowSpace(ops, Xd, n, gp, A.TauMode.Median)
def t(f, reps=7):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); out.append((time.perf_counter() - t0) * 1e3)
    return sorted(out)[len(out) // 2], r
ms_g, _ = t(lambda: sas.build_item_graph(), 3)
ms_all, _ = t(lambda: sas.build_lambdas(cent))
ms_f, gl = t(lambda: ops.feature_laplacian(cent, gp, n))
ms_l, lam = t(lambda: ops.lambdas(Xd, gl, A.TauMode.Median))
lamt = lam if torch.is_tensor(lam) else torch.from_numpy(np.asarray(lam)).cuda()
ms_s, _ = t(lambda: torch.stack([lamt.min(), -lamt.max(), lamt.sum()]).cpu())
def mk():
    sas.build_item_graph()
    torch.cuda.synchronize(); t0 = time.perf_counter(); ix = ops.make_index(Xd, lam, A.TauMode.Median, 0); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
ms_i = sorted(mk() for _ in range(5))[2]
print("item graph %.3f ms | build_lambdas %.3f ms = feature graph %.3f + lambdas %.3f + statistics %.3f + index (behind a build) %.3f" % (ms_g, ms_all, ms_f, ms_l, ms_s, ms_i))
