"""Dev timing: the scoring kernel of the re-rank with parts of it switched off (debug 64 + (bits << 8): 1 = chains of 16 values,
2 = no ranking, 4 = no chains; outputs undefined).  usage: rerank_phases.py [n]"""
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
for bits in (0, 1, 2, 4, 6, 0):
    ctx.set_option("debug", (64 + (bits << 8)) if bits else 0)
    for it in range(2):
        ctx.profile(True)
        try:
            A.build_laplacian_matrix(Xd, gp, ctx=ctx)
        except Exception as e:
            pass
        ctx.synchronize()
        prof = ctx.profile_all(); ctx.profile(False)
    print("bits %d: knn_rerank %.2f ms" % (bits, prof["knn_rerank"][0]), flush=True)
