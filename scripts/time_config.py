#!/usr/bin/env python3
"""Item graph of one configuration: wall time and every profiled stage.  usage: time_config.py n d [reps] [rank] [clusters] [opt=value ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch  # noqa: E402

import arrowspace_amd as A  # noqa: E402
from arrowspace_amd import synth  # noqa: E402

n, d = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rank = int(sys.argv[4]) if len(sys.argv) > 4 else 8
clusters = int(sys.argv[5]) if len(sys.argv) > 5 else 256
ctx = A.default_context()
seed = 42
for kv in sys.argv[6:]:
    k, v = kv.split("=")
    if k == "seed":
        seed = int(v)
    else:
        ctx.set_option(k, int(v))
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
Xd, cent = synth.mixture_device(n, d, "cuda", seed=seed, n_clusters=clusters, rank=rank)
A.build_laplacian_matrix(Xd, gp, ctx=ctx)
ctx.synchronize()
ctx.profile(True)
t0 = time.perf_counter()
each = []
for _ in range(reps):
    t1 = time.perf_counter()
    gl = A.build_laplacian_matrix(Xd, gp, ctx=ctx)
    ctx.synchronize()
    each.append((round((time.perf_counter() - t1) * 1e3, 2), ctx.stat("knn_join_theta_grouped"), ctx.stat("knn_join_theta_grouped_bad"), ctx.stat("knn_join_order_pivots"),
                 ctx.stat("knn_join_theta_far_rows"), ctx.stat("knn_join_theta_dip_rows"), ctx.stat("knn_join_candidates")))
dt = (time.perf_counter() - t0) / reps
print("each call (ms, grouped, bad, pivots, far, dip, candidates):", each)
prof = ctx.profile_all()
ctx.profile(False)
print("%d x %d rank %d clusters %d: build %.2f ms, nnz %d" % (n, d, rank, clusters, dt * 1e3, gl.matrix.nnz()))
for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    if v[1]:
        print("  %-36s %8.3f ms" % (k, v[0] / reps))
print("  stats:", {k: ctx.stat(k) for k in ("knn_join", "knn_join_prefilter", "knn_join_cone_tiles", "knn_join_cone_tiles_kept_e6", "knn_join_theta_grouped",
                                            "knn_join_candidates", "knn_rerank_scored", "knn_rerank_pairs_scored", "knn_rerank_batches",
                                            "knn_uncertified_rows", "knn_uncertified_rows_stage1a", "knn_join_blocks", "knn_join_overflow_rows", "knn_join_max_per_row",
                                            "knn_join_theta_grouped_bad", "knn_join_order_chain", "knn_rerank_open_list_ran_over", "knn_rerank_open_forced",
                                            "knn_rerank_open_undelivered", "knn_rerank_open_too_few", "knn_rerank_open_kth_not_clear", "knn_join_order_assign_off",
                                            "knn_join_order_pivots", "knn_join_order_cover_rounds", "knn_join_order_cover_low_e3", "knn_join_order_cover_left_e3", "knn_join_order_far_rows", "knn_join_theta_far_rows", "knn_join_theta_dip_rows", "knn_join_cone_split", "knn_join_cone_tiles_kept_whole_e6", "knn_join_order_cluster_rows", "knn_join_theta_band_tiles", "order_tile_labels_q10", "order_tile_labels_q50", "order_tile_labels_q90", "order_tile_major_share_e3")})
# how many times the host waited for the stream to read something back, by stage (counted over the profiled calls, per call here)
reads = {k: ctx.stat("host_reads_in_" + k) for k in list(prof.keys()) + ["no_stage"]}
print("  host read-backs per call:", {k: round(v / reps, 1) for k, v in sorted(reads.items(), key=lambda kv: -kv[1]) if v > 0})
