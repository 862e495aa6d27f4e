"""Experiment: the masked f16 join at dim 64 with one or two workgroups per CU (debug bits 0x1000000: grid 512, 0x2000000: 24 KB of
dynamic LDS padding = one workgroup per CU)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n, d = 1000000, int(sys.argv[1]) if len(sys.argv) > 1 else 64
Xd, cent = synth.mixture_device(n, d, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
A.build_laplacian_matrix(Xd, gp, ctx=ctx); ctx.synchronize()
for dbg in (0, 0x2000000, 0x1000000, 0x3000000, 0, 0x1000000):
    ctx.set_option("debug", dbg)
    ts = []
    for it in range(3):
        ctx.profile(True)
        A.build_laplacian_matrix(Xd, gp, ctx=ctx)
        ctx.synchronize()
        prof = ctx.profile_all(); ctx.profile(False)
        ts.append(prof["knn_mfma_join_f16"][0])
    print("debug=%#x: join_f16 %s ms, blocks %d listed %d cand/row %.1f" % (dbg, ["%.3f" % t for t in ts], ctx.stat("knn_mfma_join_f16_blocks"),
          ctx.stat("knn_mfma_join_f16_blocks_listed"), ctx.stat("knn_join_candidates") / n), flush=True)
ctx.set_option("debug", 0)
