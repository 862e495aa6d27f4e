import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
ref = None
for slack in (8, 6, 0):
    ctx.set_option("cand_slack", slack)
    for it in range(2):
        ctx.profile(True)
        ctx.check(ctx.lib.as_knn_cosine(ctx.h, Xd.data_ptr(), n, 128, 0, n, C.byref(c), idx.data_ptr(), dst.data_ptr()))
        torch.cuda.synchronize()
    prof = ctx.profile_all()
    ms, cnt = prof["knn_mfma_candidates_f32"]
    tot = sum(v[0] for v in prof.values())
    print("n=%d slack=%d KC=%d: candidates %.2f ms (%.1f TF/s) all kernels %.2f ms; uncertified %d -> stage2 %d" % (
        n, slack, ctx.stat("knn_cand_capacity"), ms / cnt, 2.0 * n * n * 128 / (ms / cnt * 1e-3) / 1e12, tot,
        ctx.stat("knn_uncertified_rows"), ctx.stat("knn_uncertified_rows_stage2")))
    print("   ", {k: round(v[0], 2) for k, v in prof.items()})
    cur = (idx.clone(), dst.clone())
    if ref is None: ref = cur
    else: print("    identical:", bool((ref[0] == cur[0]).all() and (ref[1] == cur[1]).all()))
