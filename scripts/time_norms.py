"""Dev timing + check of row_norms at n x 128 (the staged 64-row form against numpy, bit for bit)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = 1000000
Xd, cent = synth.mixture_device(n, 128, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
for it in range(3):
    ctx.profile(True)
    gl = A.build_laplacian_matrix(Xd, gp, ctx=ctx); ctx.synchronize()
    prof = ctx.profile_all(); ctx.profile(False)
print("row_norms %.3f ms, basis %.3f, pack %.3f" % (prof["row_norms"][0], prof["knn_join_prefilter_basis"][0], prof["pack_unit_rows"][0]), "nnz", gl.matrix.nnz())
