"""A/B of the f16 image of wide rows: one coalesced pass per (tile, 128 columns) against the per-fragment-lane packer + error pass.
The item graph must be the same (it is exact either way); the stage time is what changes.  usage: check_wide_pack.py [n] [d]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

import arrowspace_amd as A  # noqa: E402
from arrowspace_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
Xd, _ = synth.mixture_device(n, d, "cuda", seed=3, n_clusters=256, rank=8)
out = {}
for opt in (0, 1):
    ctx = A.Context()
    ctx.set_option("join_pack_wide_tiles", opt)
    A.build_laplacian_matrix(Xd, gp, ctx=ctx)
    ctx.synchronize()
    ctx.profile(True)
    t0 = time.perf_counter()
    gl = A.build_laplacian_matrix(Xd, gp, ctx=ctx)
    ctx.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    prof = ctx.profile_all()
    m = gl.matrix
    out[opt] = (m.indptr().copy(), m.indices().copy(), m.data().copy())
    print("tiles=%d: build %.2f ms, pack %.3f ms, candidates %d, uncertified %d" % (opt, ms, prof.get("pack_unit_rows", (0, 0))[0], ctx.stat("knn_join_candidates"), ctx.stat("knn_uncertified_rows")))
same = all(np.array_equal(a, b) for a, b in zip(out[0], out[1]))
print("same graph:", same)
sys.exit(0 if same else 1)
