"""Dev script: timing + certification statistics of the MFMA neighbour search on MI355X."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    ref = None
    for mode in ([32, 64, 0] if n <= 20000 else [32, 64]):
        ctx.set_option("knn_mfma", int(mode != 0)); ctx.set_option("cand_precision", mode or 32)
        for it in range(2):
            ctx.profile(True)
            torch.cuda.synchronize(); t0 = time.time()
            gl = A.build_laplacian_matrix(Xd, gp, ctx=ctx)
            torch.cuda.synchronize(); dt = time.time() - t0
        prof = ctx.profile_all()
        print("n=%d mode=%s build %.3fs nnz=%d uncert=%d stage2=%d KC=%d splits=%d" % (
            n, mode, dt, gl.matrix.nnz(), ctx.stat("knn_uncertified_rows"), ctx.stat("knn_uncertified_rows_stage2"),
            ctx.stat("knn_cand_capacity"), ctx.stat("knn_col_splits")))
        for k, (ms, cnt) in sorted(prof.items()):
            print("   %-28s %10.3f ms /%d" % (k, ms, cnt))
        for nm in ("knn_mfma_candidates_f32", "knn_mfma_candidates"):
            if nm in prof and prof[nm][1]:
                print("   %s: %.2f TFLOP/s" % (nm, 2.0 * n * n * 128 / (prof[nm][0] / prof[nm][1] * 1e-3) / 1e12))
        got = gl.matrix.to_host()
        if ref is None: ref = got
        else: print("   identical to first mode:", all(np.array_equal(a, b) for a, b in zip(ref, got)))
