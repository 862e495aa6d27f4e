"""A / B of one option over the bench's geometry sweep (1 M x 128 mixtures of rank 8 .. 128): item-graph ms per setting.
usage: ab_geometry.py <option> <a> <b> [<c>]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
import arrowspace_amd as A
from arrowspace_amd import synth
opt, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
ctx = A.default_context()
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
for rk, cl in ((8, 256), (16, 1024), (32, 2048), (64, 4096), (128, 4096)):
    Xd, _ = synth.mixture_device(1_000_000, 128, "cuda", seed=43 if rk > 8 else 42, n_clusters=cl, rank=rk)
    out = []
    for v in vals:
        ctx.set_option(opt, v)
        A.build_laplacian_matrix(Xd, gp, ctx=ctx); ctx.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); gl = A.build_laplacian_matrix(Xd, gp, ctx=ctx); ctx.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        out.append("%s=%d: %.2f ms (nnz %d, tail %.3f, cand/row %.1f)" % (opt, v, sorted(ts)[1], gl.matrix.nnz(), ctx.stat("knn_join_prefilter_tail_e6") * 1e-6,
                                                                            ctx.stat("knn_join_candidates") / 1e6))
        del gl
    print("rank %3d / %4d clusters:  " % (rk, cl) + "   ".join(out), flush=True)
    del Xd
