"""Dev script: timing + certification statistics of the MFMA neighbour search on MI355X."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth

ctx = A.default_context()
sizes = [int(x) for x in sys.argv[1:]] or [20000, 100000]
for n in sizes:
    X, cent = synth.mixture(n, 128, seed=42)
    Xd = torch.from_numpy(X).cuda()
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    for mode in ([1, 0] if n <= 20000 else [1]):
        ctx.set_option("knn_mfma", mode)
        ctx.profile(True)
        for it in range(2):
            torch.cuda.synchronize(); t0 = time.time()
            gl = A.build_laplacian_matrix(Xd, gp, ctx=ctx)
            torch.cuda.synchronize(); dt = time.time() - t0
        prof = ctx.profile_all()
        print("n=%d mode=%s build %.3fs nnz=%d uncert=%d KC=%d splits=%d" % (
            n, "mfma" if mode else "exact", dt, gl.matrix.nnz(), ctx.stat("knn_uncertified_rows"),
            ctx.stat("knn_cand_capacity"), ctx.stat("knn_col_splits")))
        for k, (ms, cnt) in sorted(prof.items()):
            print("   %-24s %10.3f ms /%d" % (k, ms / max(cnt, 1), cnt))
        if mode == 1:
            ms, cnt = prof.get("knn_mfma_candidates", (0, 1))
            fl = 2.0 * n * n * 128
            print("   candidates: %.2f TFLOP/s f64 (n_pad flops %.2f)" % (fl / (ms / cnt * 1e-3) / 1e12, 0))
            ref = gl.matrix.to_host()
        else:
            got = gl.matrix.to_host()
            print("   exact == mfma:", all(np.array_equal(a, b) for a, b in zip(ref, got)))
