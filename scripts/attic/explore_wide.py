import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n, d = int(sys.argv[1]), int(sys.argv[2])
X, cent = synth.mixture(n, d, seed=42)
Xd = torch.from_numpy(X).cuda()
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
for it in range(2):
    ctx.profile(True)
    torch.cuda.synchronize(); t0 = time.time()
    gl = A.build_laplacian_matrix(Xd, gp, ctx=ctx)
    torch.cuda.synchronize(); dt = time.time() - t0
prof = ctx.profile_all()
print("n=%d d=%d build %.3fs nnz=%d uncert=%d" % (n, d, dt, gl.matrix.nnz(), ctx.stat("knn_uncertified_rows")))
for k, (ms, cnt) in sorted(prof.items()):
    print("   %-28s %10.3f ms /%d" % (k, ms, cnt))
ms, cnt = prof["knn_mfma_candidates_f32"]
print("   f32 candidates: %.1f TFLOP/s" % (2.0 * n * n * d / (ms / cnt * 1e-3) / 1e12))
