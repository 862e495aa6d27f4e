"""Grouped thresholds against the table size: how many of the 4 096 checked rows sit below the plain sample's threshold."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
for name, val in [a.split("=") for a in sys.argv[1:]]:
    ctx.set_option(name, int(val))
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
for n in (1_000_000, 2_000_000, 3_000_000, 4_000_000):
    for (rank, clusters) in ((8, 256), (64, 4096)):
        Xh, _ = synth.mixture_device(n, 128, "cuda", seed=43, n_clusters=clusters, rank=rank)
        A.build_laplacian_matrix(Xh, gp, ctx=ctx)
        ctx.profile(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gl = A.build_laplacian_matrix(Xh, gp, ctx=ctx)
        ctx.synchronize(); dt = time.perf_counter() - t0
        pr = ctx.profile_all(); ctx.profile(False)
        print("n %d rank %d clusters %d build %.1f ms thresholds %.1f ms" % (n, rank, clusters, dt * 1e3, pr["knn_join_thresholds"][0]),
              {k: ctx.stat(k) for k in ("knn_join_theta_grouped", "knn_join_theta_grouped_bad", "knn_join_candidates")}, flush=True)
        del gl, Xh
