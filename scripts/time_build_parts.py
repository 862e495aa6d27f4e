"""Host-side split of the bench's build step (1 M x 128): item graph, feature graph, lambdas, statistics, index."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
import arrowspace_amd as A
from arrowspace_amd import distributed as D
from arrowspace_amd import synth
n, d = 1_000_000, 128
ctx = A.default_context()
ops = D.HipOps(ctx)
Xd, cent = synth.mixture_device(n, d, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
def T(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t0) * 1e3
for rep in range(4):
    gl_items, t_graph = T(lambda: A.build_laplacian_matrix(Xd, gp, ctx=ctx))
    gl, t_feat = T(lambda: ops.feature_laplacian(cent, gp, n))
    lam, t_lam = T(lambda: ops.lambdas(Xd, gl, A.TauMode.Median))
    st, t_stats = T(lambda: (float(lam.min()), float(lam.max()), float(lam.sum())))
    ix, t_index = T(lambda: ops.make_index(Xd, lam, A.TauMode.Median, 0))
    _, t_free = T(lambda: None)
    t0 = time.perf_counter(); del gl_items; torch.cuda.synchronize(); t_del = (time.perf_counter() - t0) * 1e3
    print("item graph %.2f  feature graph %.2f  lambdas %.2f  statistics %.2f  index %.2f  freeing the graph %.2f ms" % (t_graph, t_feat, t_lam, t_stats, t_index, t_del), flush=True)
    del ix, gl
