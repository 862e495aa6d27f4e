"""Dev check of the threshold scan of the search (stage 0 of K9): identical results with and without it, failures, timing."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import distributed as D
from arrowspace_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
cfgs = [tuple(int(v) for v in x.split(":")) for x in sys.argv[4:]] or [(32, 3, 512)]
ctx = A.Context(0)
ops = D.HipOps(ctx)
X, cent = synth.mixture(n, d, seed=42)
Xd = torch.from_numpy(X).cuda()
qidx = synth.query_indices(n, nq, seed=42)
Q = torch.from_numpy(synth.mixture_at(qidx, d, seed=42) * 1.02).cuda()
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
sas = D.ShardedArrowSpace(ops, Xd, n, gp, A.TauMode.Median)
sas.build_item_graph(); sas.build_lambdas(cent)
lq = sas.prepare_queries(Q)
res = {}
for scan, rate, m, cap in [(0, 32, 3, 512)] + [(1,) + c for c in cfgs]:
    ctx.set_option("search_scan", scan); ctx.set_option("search_scan_sample", rate); ctx.set_option("search_scan_rank", m); ctx.set_option("search_scan_cap", cap)
    sas._index = None if hasattr(sas, "_index") else None
    for rep in range(3):
        ctx.profile(True)
        torch.cuda.synchronize(); t0 = time.time()
        idx, sc = sas.search(Q, lq, 10, 0.7)
        torch.cuda.synchronize(); t1 = time.time()
        prof = ctx.profile_all()
    res[scan] = (idx.cpu().numpy(), sc.cpu().numpy())
    print("scan=%d rate %d m %d cap %d: %.2f ms  uncertified %d (stage2 %d)  cand/query %.1f max %d overflow %d" % (
        scan, rate, m, cap, (t1 - t0) * 1e3, ctx.stat("search_uncertified_rows"), ctx.stat("search_uncertified_rows_stage2"),
        ctx.stat("search_scan_candidates") / nq if scan else 0, ctx.stat("search_scan_max_per_query") if scan else 0,
        ctx.stat("search_scan_overflow_queries") if scan else 0), {k: round(v[0] / v[1], 2) for k, v in prof.items()})
    if scan:
        b = ctx.stat("search_scan_bf16_blocks"); f = ctx.stat("search_scan_bf16_blocks_full")
        print("   blocks of the last scan launch: %d, in full %.3f" % (b, f / max(1, b)))
        print("   identical:", np.array_equal(res[0][0], res[1][0]), np.array_equal(res[0][1], res[1][1]))
