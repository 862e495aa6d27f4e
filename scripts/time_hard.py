"""Profile groups of the item graph on the bench's hard mixture (rank 64 / 4 096 clusters): where the 147 ms go."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
import arrowspace_amd as A
from arrowspace_amd import synth
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 128
ctx = A.default_context()
rank, clusters, kind, eps = 64, 4096, "mixture", 0.5
for name, val in [a.split("=") for a in sys.argv[2:]]:
    if name == "rank": rank = int(val)
    elif name == "kind": kind = val          # mixture | normal | dup (every row 16 times)
    elif name == "eps": eps = float(val)
    elif name == "d": d = int(val)
    elif name == "clusters": clusters = int(val)
    elif name in ("kind", "eps", "d"): pass
    else: ctx.set_option(name, int(val))
if kind == "normal":
    Xh = torch.randn((n, d), dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7))
elif kind == "dup":
    base, _ = synth.mixture_device(n // 16, d, "cuda", seed=43, n_clusters=clusters, rank=rank)
    Xh = base.repeat_interleave(16, dim=0)[torch.randperm(n // 16 * 16, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))].contiguous()
    n = Xh.shape[0]
else:
    Xh, _ = synth.mixture_device(n, d, "cuda", seed=43, n_clusters=clusters, rank=rank)
gp = A.GraphParams(eps, 16, 16, 2.0, 0.25)
A.build_laplacian_matrix(Xh, gp, ctx=ctx)
for rep in range(2):
    ctx.profile(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gl = A.build_laplacian_matrix(Xh, gp, ctx=ctx)
    ctx.synchronize(); dt = time.perf_counter() - t0
    pr = ctx.profile_all(); ctx.profile(False)
    print("rank %d clusters %d tail %.3f build %.1f ms nnz %d" % (rank, clusters, ctx.stat("knn_join_prefilter_tail_e6") * 1e-6, dt * 1e3, gl.matrix.nnz()), {k: (round(v[0], 2), v[1]) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0])},
          {k: ctx.stat(k) for k in ("knn_join_prefilter_rotated", "knn_join_row_order", "knn_join_theta_grouped", "knn_uncertified_rows", "knn_join_candidates")}, flush=True)
    del gl
