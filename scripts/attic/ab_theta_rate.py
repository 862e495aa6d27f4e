"""Dev timing: the build under denser grouped threshold samples and other ranks.  usage: ab_theta_rate.py [n]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
Xd, cent = synth.mixture_device(n, 128, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
A.build_laplacian_matrix(Xd, gp, ctx=ctx); ctx.synchronize()
ref = None
for rate, rank in ((32, 4), (16, 4), (16, 6), (8, 4), (8, 6), (8, 8), (4, 8), (4, 12), (32, 4)):
    ctx.set_option("join_theta_rate", rate); ctx.set_option("join_rank", rank)
    for it in range(2):
        ctx.profile(True)
        gl = A.build_laplacian_matrix(Xd, gp, ctx=ctx)
        ctx.synchronize()
        prof = ctx.profile_all(); ctx.profile(False)
    nnz = gl.matrix.nnz()
    ref = nnz if ref is None else ref
    tot = sum(v[0] for k, v in prof.items() if k.startswith("knn_") or k in ("pack_unit_rows", "row_norms", "graph_assembly"))
    print("rate 1/%d rank %d: kernels %.2f ms | thresholds %.2f cone %.2f first %.2f f16 %.2f second %.2f rerank %.2f | cand/row %.1f open %d -> %d grouped %d nnz %s" % (
        rate, rank, tot, prof["knn_join_thresholds"][0], prof.get("knn_join_cone_tiles", (0, 0))[0], prof["knn_join_prefilter"][0], prof["knn_mfma_join_f16"][0],
        prof.get("knn_mfma_join_f16_second", (0, 0))[0], prof["knn_rerank"][0], ctx.stat("knn_join_candidates") / n, ctx.stat("knn_uncertified_rows"),
        ctx.stat("knn_uncertified_rows_stage1a"), ctx.stat("knn_join_theta_grouped"), "same" if nnz == ref else "DIFFERENT"), flush=True)
