"""Dev check of the ordered search scan and its int8 first pass (items up to 128 wide): identical results with and without,
stage times.  usage: explore_search_prefilter.py [n] [d] [nq]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
Xd, cent = synth.mixture_device(n, d, "cuda", seed=42)
g = torch.Generator(device="cuda"); g.manual_seed(7)
lam = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 0.3 + 0.05
qi = torch.randint(0, n, (nq,), device="cuda", generator=g)
Q = (Xd[qi] * 1.01 + 0.02 * torch.randn(nq, d, dtype=torch.float64, device="cuda", generator=g)).contiguous()
lq = (lam[qi] * 1.02).contiguous()
res = {}
if os.environ.get("JOIN_BLOCK"): ctx.set_option("join_block", int(os.environ["JOIN_BLOCK"]))
for order, pre in ((0, 0), (2, 0), (2, 1)):
    ctx.set_option("search_scan_order", order); ctx.set_option("search_prefilter", pre)
    asp = A.ArrowSpace(Xd, A.TauMode.Median, ctx=ctx)  # (a fresh index: the images are kept with it)
    asp.update_lambdas(lam)
    for rep in range(3):
        ctx.profile(True)
        torch.cuda.synchronize(); t0 = time.time()
        idx, sc = asp.search_lambda_aware_batch(Q, lq, 10, 0.7)
        torch.cuda.synchronize(); t1 = time.time()
        prof = ctx.profile_all()
    res[(order, pre)] = (idx.cpu().numpy() if torch.is_tensor(idx) else np.asarray(idx), sc.cpu().numpy() if torch.is_tensor(sc) else np.asarray(sc))
    print("order=%d prefilter=%d: %.2f ms -> %.0f QPS; used order %d prefilter %d; blocks %d flagged %d; uncertified %d" % (
        order, pre, (t1 - t0) * 1e3, nq / (t1 - t0), ctx.stat("search_scan_order"), ctx.stat("search_scan_prefilter"),
        ctx.stat("search_scan_prefilter_blocks"), ctx.stat("search_scan_prefilter_blocks_flagged"), ctx.stat("search_scan_uncertified_first")),
        {k: round(v[0] / max(1, v[1]), 2) for k, v in prof.items()}, flush=True)
ks = list(res)
for k in ks[1:]:
    print("identical to the unordered scan:", k, np.array_equal(res[ks[0]][0], res[k][0]), np.array_equal(res[ks[0]][1], res[k][1]))

# small batches: the streaming scan with and without the int8 filter (the index has been searched before)
for b in (1, 8, 32):
    for flt in (0, 1):
        ctx.set_option("search_stream_filter", flt)
        Qb, lb = Q[:b].contiguous(), lq[:b].contiguous()
        for rep in range(3): out = asp.search_lambda_aware_batch(Qb, lb, 10, 0.7)
        ctx.profile(True)
        out = asp.search_lambda_aware_batch(Qb, lb, 10, 0.7)
        prof = ctx.profile_all()
        print("   ", {k: round(v[0], 3) for k, v in prof.items()})
        ctx.profile(False)
        torch.cuda.synchronize(); t0 = time.time()
        for rep in range(20): out = asp.search_lambda_aware_batch(Qb, lb, 10, 0.7)
        torch.cuda.synchronize(); t1 = time.time()
        oi = out[0].cpu().numpy() if torch.is_tensor(out[0]) else np.asarray(out[0])
        osc = out[1].cpu().numpy() if torch.is_tensor(out[1]) else np.asarray(out[1])
        print("nq=%d filter=%d: %.3f ms per batch (stream %d filter %d items kept %d overflow %d retries %d)  same as the big batch: %s %s" % (
            b, flt, (t1 - t0) / 20 * 1e3, ctx.stat("search_stream_scan"), ctx.stat("stream_scan_filter"), ctx.stat("stream_scan_filter_items"),
            ctx.stat("stream_scan_filter_overflow"), ctx.stat("stream_scan_retries"), np.array_equal(oi, res[ks[0]][0][:b]),
            np.array_equal(osc, res[ks[0]][1][:b])), flush=True)
ctx.set_option("search_stream_filter", 1)
