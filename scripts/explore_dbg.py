import sys, time, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
X, cent = synth.mixture(n, 128, seed=42)
Xd = torch.from_numpy(X).cuda()
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
K = gp.topk + 1
idx = torch.empty((n, K), dtype=torch.int64, device="cuda"); dst = torch.empty((n, K), dtype=torch.float64, device="cuda")
c = gp.to_c()
for wps, dbg in ((2, 0), (2, 2), (1, 0)):
    ctx.set_option("waves_per_simd", wps)
    ctx.set_option("debug", dbg)
    for it in range(2):
        ctx.profile(True)
        ctx.check(ctx.lib.as_knn_cosine(ctx.h, Xd.data_ptr(), n, 128, 0, n, C.byref(c), idx.data_ptr(), dst.data_ptr()))
        torch.cuda.synchronize()
    prof = ctx.profile_all()
    ms, cnt = prof["knn_mfma_candidates_f32"]
    print("wps=%d " % wps + "n=%d dbg=%d candidates %.2f ms  %.1f TFLOP/s" % (n, dbg, ms / cnt, 2.0 * n * n * 128 / (ms / cnt * 1e-3) / 1e12))
ctx.set_option("debug", 0)
