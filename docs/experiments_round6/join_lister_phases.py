"""The masked f16 join with lister waves at n x 128: whole, tiles consumed but not listed (debug 4), nothing dumped (debug 2).
ARROWSPACE_ALLOW_ABLATION=1."""
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
for dbg in (0, 4, 2, 0):
    ctx.set_option("debug", dbg)
    ts = []
    for it in range(3):
        ctx.profile(True)
        A.build_laplacian_matrix(Xd, gp, ctx=ctx)
        ctx.synchronize()
        prof = ctx.profile_all(); ctx.profile(False)
        ts.append(prof["knn_mfma_join_f16"][0])
    print("debug=%d lister=%d deep=%d: join_f16 %s ms, blocks %d dumped %d listed %d" % (dbg, ctx.stat("knn_join_lister"), ctx.stat("knn_join_deep"), ["%.3f" % t for t in ts],
          ctx.stat("knn_mfma_join_f16_blocks"), ctx.stat("knn_mfma_join_f16_blocks_pretest"), ctx.stat("knn_mfma_join_f16_blocks_listed")), flush=True)
ctx.set_option("debug", 0)
