import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
X, cent = synth.mixture(n, 128, seed=42)
Xd = torch.from_numpy(X).cuda()
A._adopt_torch_stream(ctx) if hasattr(A, "_adopt_torch_stream") else None
gl = A.GraphFactory.build_laplacian_matrix_from_k_cluster(cent, 0.5, 16, 16, 2.0, 0.25, False, False, n, ctx=ctx)
print("F x F nnz", gl.matrix.nnz())
out = torch.empty(n, dtype=torch.float64, device="cuda")
def run(name, fn):
    for it in range(3):
        ctx.profile(True); fn(); torch.cuda.synchronize()
    prof = ctx.profile_all()
    for k, (ms, cnt) in prof.items():
        print("  %-34s %-16s %8.3f ms   %.2f TB/s of rows" % (name, k, ms / cnt, n * 128 * 8 / (ms / cnt * 1e-3) / 1e12))
lib = ctx.lib
for v2 in (1, 0):
    ctx.set_option("lambda_v2", v2)
    run("v2=%d lambda median" % v2, lambda: ctx.check(lib.as_compute_taumode(ctx.h, Xd.data_ptr(), n, 128, gl.matrix.h, 1, 0.0, out.data_ptr())))
    run("v2=%d lambda fixed" % v2, lambda: ctx.check(lib.as_compute_taumode(ctx.h, Xd.data_ptr(), n, 128, gl.matrix.h, 0, 0.3, out.data_ptr())))
    run("v2=%d lambda mean" % v2, lambda: ctx.check(lib.as_compute_taumode(ctx.h, Xd.data_ptr(), n, 128, gl.matrix.h, 2, 0.0, out.data_ptr())))
    run("v2=%d select_tau median" % v2, lambda: ctx.check(lib.as_select_tau(ctx.h, Xd.data_ptr(), n, 128, 1, 0.0, out.data_ptr())))
    run("v2=%d select_tau fixed" % v2, lambda: ctx.check(lib.as_select_tau(ctx.h, Xd.data_ptr(), n, 128, 0, 0.3, out.data_ptr())))
