"""Dev check of the streaming lambda-tau kernel (taumode_v3.hip): parity against the oracle on
ragged sizes, all tau modes, non-finite rows; then timing at n items x 128."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
from oracle import oracle_ctypes as O
ctx = A.default_context()
lib = ctx.lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
X, cent = synth.mixture(n, 128, seed=42)
gl = A.GraphFactory.build_laplacian_matrix_from_k_cluster(cent, 0.5, 16, 16, 2.0, 0.25, False, False, n, ctx=ctx)
csr = gl.matrix.to_host()
print("F x F nnz", gl.matrix.nnz())
MODES = [("fixed", 0, 0.3, O.TAU_FIXED), ("median", 1, 0.0, O.TAU_MEDIAN), ("mean", 2, 0.0, O.TAU_MEAN), ("pct95", 3, 0.95, O.TAU_PERCENTILE)]
ok = True
for m in (1, 31, 32, 33, 4097, 20000):
    Xs = X[:m].copy()
    if m >= 33:
        Xs[5, 7] = np.inf; Xs[9, 100] = -np.inf; Xs[12, :] = 0.0; Xs[20, 3:90] = np.nan
        Xs[21] *= 1e-160; Xs[22] *= 1e150
    Xd = torch.from_numpy(Xs).cuda()
    out = torch.empty(m, dtype=torch.float64, device="cuda")
    for name, mode, arg, omode in MODES:
        ref = O.taumode_lambdas(Xs, csr, omode, arg)
        for v3 in (1, 0):
            ctx.set_option("lambda_v3", v3)
            ctx.check(lib.as_compute_taumode(ctx.h, Xd.data_ptr(), m, 128, gl.matrix.h, mode, arg, out.data_ptr()))
            got = out.cpu().numpy()
            both_nan = np.isnan(ref) & np.isnan(got)
            rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
            rel[both_nan] = 0
            bad = ~(rel <= 1e-12)
            if bad.any():
                ok = False
                i = np.where(bad)[0][:5]
                print("MISMATCH m=%d %s v3=%d: %d rows, e.g." % (m, name, v3, bad.sum()), i, got[i], ref[i])
print("parity", "OK" if ok else "FAILED")
Xd = torch.from_numpy(X).cuda()
out = torch.empty(n, dtype=torch.float64, device="cuda")
def run(name, fn):
    for it in range(4):
        ctx.profile(True); fn(); torch.cuda.synchronize()
    prof = ctx.profile_all()
    for k, (ms, cnt) in prof.items():
        print("  %-28s %-16s %8.3f ms   %.2f TB/s of rows" % (name, k, ms / cnt, n * 1032 / (ms / cnt * 1e-3) / 1e12))
for v3 in (1, 0):
    ctx.set_option("lambda_v3", v3)
    for name, mode, arg, _ in MODES:
        run("v3=%d %s" % (v3, name), lambda: ctx.check(lib.as_compute_taumode(ctx.h, Xd.data_ptr(), n, 128, gl.matrix.h, mode, arg, out.data_ptr())))
ctx.set_option("lambda_v3", 1); ctx.set_option("debug", 4)
for name, mode, arg, _ in MODES:
    run("lockstep %s" % name, lambda: ctx.check(lib.as_compute_taumode(ctx.h, Xd.data_ptr(), n, 128, gl.matrix.h, mode, arg, out.data_ptr())))
ctx.set_option("debug", 3)
for name, mode, arg, _ in MODES:
    print(name); ctx.check(lib.as_compute_taumode(ctx.h, Xd.data_ptr(), n, 128, gl.matrix.h, mode, arg, out.data_ptr())); torch.cuda.synchronize()
