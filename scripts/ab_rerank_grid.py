"""Dev timing: the scoring kernel of the re-rank under different numbers of workgroups (one wave each).  usage: ab_rerank_grid.py [n]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
Xd, cent = synth.mixture_device(n, 128, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
A.build_laplacian_matrix(Xd, gp, ctx=ctx); ctx.synchronize()
for grid in (0, 1024, 1536, 1792, 2048, 2560, 4096, 8192, 0):
    ctx.set_option("rerank_grid", grid)
    ts = []
    for it in range(3):
        ctx.profile(True)
        gl = A.build_laplacian_matrix(Xd, gp, ctx=ctx)
        ctx.synchronize()
        prof = ctx.profile_all(); ctx.profile(False)
        ts.append(prof["knn_rerank"][0])
    print("rerank_grid=%d: re-rank %s ms (nnz %d)" % (grid, ["%.3f" % t for t in ts], gl.matrix.nnz()), flush=True)
