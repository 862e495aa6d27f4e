"""Dev timing of the lambda-tau pass at other widths than 128 (configs[4]: F = 768).  usage: explore_taumode.py [n] [F] [mode]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 768
mode = sys.argv[3] if len(sys.argv) > 3 else "median"
Xd, cent = synth.mixture_device(n, F, "cuda", seed=42)
import ctypes as C
from arrowspace_amd._lib import ptr
centroids = cent.cpu().numpy() if torch.is_tensor(cent) else np.asarray(cent)
gl = A.GraphFactory.build_laplacian_matrix_from_k_cluster(centroids, 0.5, 16, 16, 2.0, 0.25, False, False, n)
tm = {"median": A.TauMode.Median, "mean": A.TauMode.Mean, "p95": A.TauMode.Percentile(0.95), "fixed": A.TauMode.Fixed(0.5)}[mode]
print("feature graph: %s, nnz %d" % (gl.matrix.shape(), gl.matrix.nnz()))
out = torch.empty(n, dtype=torch.float64, device="cuda")
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    ctx.check(ctx.lib.as_compute_taumode(ctx.h, ptr(Xd), n, F, gl.matrix.h, tm.kind, tm.arg, ptr(out)))
    torch.cuda.synchronize(); t1 = time.time()
    print("n=%d F=%d %s: %.3f ms -> %.1f GB/s of n(8F+8) bytes" % (n, F, mode, (t1 - t0) * 1e3, n * (8 * F + 8) / (t1 - t0) / 1e9), flush=True)
