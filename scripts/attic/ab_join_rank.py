"""Neighbour table at 1 M x 128 against the order statistic of the join's thresholds (join_rank: the m-th best of the sample)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import ctypes as C
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 128
Xd, _ = synth.mixture_device(n, d, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
def knn():
    K = gp.topk + 1
    idx = torch.empty((n, K), dtype=torch.int64, device="cuda"); dst = torch.empty((n, K), dtype=torch.float64, device="cuda")
    A.api._adopt_torch_stream(ctx); c = gp.to_c()
    ctx.check(ctx.lib.as_knn_cosine(ctx.h, Xd.data_ptr(), n, d, 0, n, C.byref(c), idx.data_ptr(), dst.data_ptr()))
    torch.cuda.synchronize(); return idx, dst
ref = knn()
for rank in (0, 3, 2, 1, 6):
    ctx.set_option("join_rank", rank)
    knn(); ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(3): r = knn()
    dt = (time.perf_counter() - t0) / 3
    pr = ctx.profile_all(); ctx.profile(False)
    g = lambda k: pr[k][0] / 3 if k in pr else 0.0
    print("rank", rank, "table ms %.2f" % (dt * 1e3), "first %.2f f16 %.2f rerank %.2f second %.2f thresholds %.2f stream %.2f" % (
        g("knn_join_prefilter"), g("knn_mfma_join_f16"), g("knn_rerank"), g("knn_mfma_join_f16_second"), g("knn_join_thresholds"), g("knn_stream_scan")),
        "cand/row %.1f" % (ctx.stat("knn_join_candidates") / n), "uncertified", ctx.stat("knn_uncertified_rows"), "after second", ctx.stat("knn_uncertified_rows_stage1a"),
        "equal", torch.equal(ref[0], r[0]) and torch.equal(ref[1], r[1]), flush=True)
ctx.set_option("join_rank", 0)
