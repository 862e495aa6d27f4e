#!/usr/bin/env python3
"""One item graph with the re-rank scoring every pair once and with every row scoring all of its survivors ("rerank_share" 1 / 0): same graph, stage times.
usage: try_rerank_pairs.py [n] [d]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import arrowspace_amd as A  # noqa: E402
from arrowspace_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ctx = A.Context(0)
ctx.set_option("join_min_rows", 512)
X, _ = synth.mixture_device(n, d, torch.device("cuda", 0), seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
out = {}
for mode in (0, 1, 1):
    ctx.set_option("rerank_share", mode)
    ctx.profile(True)
    t0 = time.perf_counter()
    gl = A.build_laplacian_matrix(X, gp, ctx=ctx)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    pr = ctx.profile_all()
    ctx.profile(False)
    indptr, indices, data = gl.matrix.to_host()
    out[mode] = (indptr, indices, data)
    print("rerank_share=%d: %.2f ms, nnz %d, uncertified %d, scored %d, pairs %d, batches %d; %s" % (
        mode, dt * 1e3, len(indices), ctx.stat("knn_uncertified_rows"), ctx.stat("knn_rerank_scored"), ctx.stat("knn_rerank_pairs_scored"),
        ctx.stat("knn_rerank_batches"), {k: round(v[0], 3) for k, v in pr.items() if "rerank" in k}), flush=True)
same = all(np.array_equal(out[0][i], out[1][i]) for i in range(3))
print("shared and unshared scoring give the same graph:", same)
sys.exit(0 if same else 1)
