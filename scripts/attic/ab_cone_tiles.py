"""Dev timing: the join with and without the cone bound on tile pairs in front of the first pass.  usage: ab_cone_tiles.py [n] [rank] [clusters]"""
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
for cone in (0, 1, 0, 1):
    ctx.set_option("join_cone_tiles", cone)
    ctx.set_option("join_cone_tiles_max_e3", 1000)
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
    print("cone=%d: first pass %.3f ms, cone tiles %.3f ms (kept %.4f), f16 join %.3f, build kernels %.2f ms; blocks passed on %d; nnz %d, same graph: %s" % (
        cone, prof["knn_join_prefilter"][0], prof.get("knn_join_cone_tiles", (0, 0))[0], ctx.stat("knn_join_cone_tiles_kept_e6") * 1e-6,
        prof["knn_mfma_join_f16"][0], tot, ctx.stat("knn_join_prefilter_blocks_flagged"), gl.matrix.nnz(), same), flush=True)
