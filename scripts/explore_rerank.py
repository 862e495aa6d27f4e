"""Dev check of the split re-rank (rerank_split.h) against the one-kernel form: identical neighbour tables, stage times.
usage: explore_rerank.py [n] [K] [d]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
topk = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
Xd, cent = synth.mixture_device(n, d, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, topk, 2.0, 0.25)
ctx.set_option("join_min_rows", 512)
res = {}
for split in (0, 1):
    ctx.set_option("rerank_split", split)
    for rep in range(2):
        ctx.profile(True)
        torch.cuda.synchronize(); t0 = time.time()
        idx, dist = A.knn_cosine(Xd, gp, ctx=ctx)
        torch.cuda.synchronize(); t1 = time.time()
        prof = ctx.profile_all()
    res[split] = (idx.cpu().numpy() if torch.is_tensor(idx) else idx, dist.cpu().numpy() if torch.is_tensor(dist) else dist)
    print("split=%d  %.1f ms  uncertified %d" % (split, (t1 - t0) * 1e3, ctx.stat("knn_uncertified_rows")),
          {k: round(v[0] / v[1], 2) for k, v in prof.items()}, flush=True)
print("tables identical:", np.array_equal(res[0][0], res[1][0]), np.array_equal(res[0][1], res[1][1]))
