"""Dev timing: the masked f16 pass walking a list of the units that hold something (join_unit_list) against drawing every unit.
usage: ab_unit_list.py [n] [rank] [clusters]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 8
clusters = int(sys.argv[3]) if len(sys.argv) > 3 else 256
seed = 42 if (rank, clusters) == (8, 256) else 43
Xd, cent = synth.mixture_device(n, 128, "cuda", seed=seed, n_clusters=clusters, rank=rank)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
A.build_laplacian_matrix(Xd, gp, ctx=ctx); ctx.synchronize()
ref = None
for band in (0, 1, 0, 1):
    ctx.set_option("join_unit_list", band)
    for it in range(2):
        ctx.profile(True)
        gl = A.build_laplacian_matrix(Xd, gp, ctx=ctx)
        ctx.synchronize()
        prof = ctx.profile_all(); ctx.profile(False)
    csr = gl.matrix.to_host()
    if ref is None:
        ref = csr
    same = all((a == b).all() for a, b in zip(ref, csr))
    tot = sum(v[0] for k, v in prof.items() if k.startswith("knn_") or k in ("pack_unit_rows", "row_norms", "graph_assembly"))
    print("unit list %d: scored/row %.2f kernels %.2f ms | thresholds %.2f + band %.2f cone %.2f first %.2f f16 %.2f second %.2f rerank %.2f scan %.2f | cand/row %.1f kept %.4f second kept %.3f open %d -> %d same graph %s" % (
        band, ctx.stat("knn_rerank_scored") / n, tot, prof["knn_join_thresholds"][0], prof.get("knn_join_thresholds_band", (0, 0))[0], prof.get("knn_join_cone_tiles", (0, 0))[0], prof["knn_join_prefilter"][0],
        prof["knn_mfma_join_f16"][0], prof.get("knn_mfma_join_f16_second", (0, 0))[0], prof["knn_rerank"][0], prof.get("knn_stream_scan", (0, 0))[0],
        ctx.stat("knn_join_candidates") / n, ctx.stat("knn_join_cone_tiles_kept_e6") * 1e-6, ctx.stat("knn_join_cone_second_kept_e6") * 1e-6, ctx.stat("knn_uncertified_rows"), ctx.stat("knn_uncertified_rows_stage1a"), same), flush=True)
