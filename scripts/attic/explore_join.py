"""Dev check of the symmetric threshold join (mfma_join_f32.hip): identical neighbour tables with and
without it, uncertified-row counts, timing."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
X, cent = synth.mixture(n, d, seed=42)
Xd = torch.from_numpy(X).cuda()
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
ctx.set_option("join_min_rows", 512)
if os.environ.get("JOIN_PIPE"): ctx.set_option("join_pipe", int(os.environ["JOIN_PIPE"]))
if os.environ.get("JOIN_BLOCK"): ctx.set_option("join_block", int(os.environ["JOIN_BLOCK"]))
if os.environ.get("AS_DEBUG"): ctx.set_option("debug", int(os.environ["AS_DEBUG"]))
res = {}
CFGS = [(0, 16, 5, 192)] + [(1, int(a), int(b), int(c)) for a, b, c in (x.split(":") for x in sys.argv[3:])] if len(sys.argv) > 3 else [(0, 16, 5, 192), (1, 16, 5, 192)]
for join, rate, m, cap in CFGS:
    ctx.set_option("knn_join", join); ctx.set_option("join_sample", rate); ctx.set_option("join_rank", m); ctx.set_option("join_cap", cap)
    for rep in range(2):
        ctx.profile(True)
        torch.cuda.synchronize(); t0 = time.time()
        idx, dist = A.knn_cosine(Xd, gp, ctx=ctx)
        torch.cuda.synchronize(); t1 = time.time()
        prof = ctx.profile_all()
    if join: print("   rate %d m %d cap %d: cand/row %.1f max %d overflow %d" % (rate, m, cap, ctx.stat("knn_join_candidates") / n, ctx.stat("knn_join_max_per_row"), ctx.stat("knn_join_overflow_rows")))
    res[join] = (idx.cpu().numpy() if torch.is_tensor(idx) else idx, dist.cpu().numpy() if torch.is_tensor(dist) else dist)
    print("join=%d  %.1f ms  uncertified %d stage1b %d stage2 %d  join_used %d" % (join, (t1 - t0) * 1e3, ctx.stat("knn_uncertified_rows"), ctx.stat("knn_uncertified_rows_stage1b"),
          ctx.stat("knn_uncertified_rows_stage2"), ctx.stat("knn_join")), {k: round(v[0] / v[1], 2) for k, v in prof.items()})
same_i = np.array_equal(res[0][0], res[1][0]); same_d = np.array_equal(res[0][1], res[1][1])
print("identical tables:", same_i, same_d)
if not same_i:
    bad = np.where((res[0][0] != res[1][0]).any(axis=1))[0]
    print(len(bad), "rows differ, e.g.", bad[:5]); print(res[0][0][bad[0]], res[1][0][bad[0]])
