"""Dev: the join statistics a build of the item graph leaves in the context. usage: print_build_stats.py [n] [rank] [clusters]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
A.build_laplacian_matrix(Xd, gp, ctx=ctx); ctx.synchronize()
for k in ("knn_join_cap", "knn_join_candidates", "knn_join_max_per_row", "knn_join_overflow_rows", "knn_rerank_scored", "knn_uncertified_rows",
          "knn_uncertified_rows_stage1a", "knn_join_cone_tiles_kept_e6", "knn_join_prefilter_blocks", "knn_join_prefilter_blocks_flagged",
          "knn_join_cone_second", "knn_join_cone_second_kept_e6", "knn_join_theta_grouped", "knn_join_theta_grouped_bad", "knn_join_row_order"):
    print(k, ctx.stat(k))
