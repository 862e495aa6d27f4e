"""Cycle stamps of the masked f16 pass (debug bit 8), lists per row against lists per group: usage stamps_join.py [n]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
Xd, cent = synth.mixture_device(n, 128, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
for grouped in (0, 1, 0, 1):
    ctx.set_option("join_grouped_lists", grouped)
    ctx.set_option("debug", 0)
    A.build_laplacian_matrix(Xd, gp, ctx=ctx); ctx.synchronize()
    ctx.profile(True)
    A.build_laplacian_matrix(Xd, gp, ctx=ctx); ctx.synchronize()
    prof = ctx.profile_all(); ctx.profile(False)
    print("grouped=%d: join_f16 %.3f ms, rerank %.3f ms, second %.3f ms, uncertified %d, listed-blocks %d" % (
        grouped, prof["knn_mfma_join_f16"][0], prof["knn_rerank"][0], prof.get("knn_mfma_join_f16_second", (0, 0))[0],
        ctx.stat("knn_uncertified_rows"), ctx.stat("knn_mfma_join_f16_blocks_listed")), flush=True)
    ctx.set_option("debug", 8)
    A.build_laplacian_matrix(Xd, gp, ctx=ctx); ctx.synchronize()
ctx.set_option("debug", 0)
