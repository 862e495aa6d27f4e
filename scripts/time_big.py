"""Dev timing: neighbour tables of the big configurations (4 M x 128, 1 M x 768) with and without the cone bound on tile pairs.
usage: time_big.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
for n, d in ((4_000_000, 128), (1_000_000, 768)):
    Xd, cent = synth.mixture_device(n, d, "cuda", seed=42)
    ref = None
    for cone in (1, 0, 1):
        ctx.set_option("join_cone_tiles", cone)
        for it in range(2):
            ctx.profile(True)
            t0 = time.perf_counter()
            gl = A.build_laplacian_matrix(Xd, gp, ctx=ctx)
            ctx.synchronize()
            dt = time.perf_counter() - t0
            prof = ctx.profile_all(); ctx.profile(False)
        nnz = gl.matrix.nnz()
        if ref is None:
            ref = nnz
        print("%d x %d cone=%d: build %.1f ms; first pass %.2f, cone tiles %.2f (kept %.4f), f16 join %.2f, re-rank %.2f; nnz %d (%s)" % (
            n, d, cone, dt * 1e3, prof["knn_join_prefilter"][0], prof.get("knn_join_cone_tiles", (0, 0))[0],
            ctx.stat("knn_join_cone_tiles_kept_e6") * 1e-6, prof["knn_mfma_join_f16"][0], prof["knn_rerank"][0], nnz, "same" if nnz == ref else "DIFFERENT"), flush=True)
        del gl
    del Xd
    torch.cuda.empty_cache()
ctx.set_option("join_cone_tiles", 1)
