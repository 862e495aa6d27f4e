"""Dev sweep of the join's threshold choice (rank m of a 1-in-s sample): total neighbour-table time and stage split.
usage: explore_thresholds.py [n] [d]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
Xd, cent = synth.mixture_device(n, d, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
import os
pairs = [tuple(int(v) for v in x.split(':')) for x in os.environ.get('PAIRS', '4:32,3:32,3:24,2:16,4:48,5:48,6:64,3:48').split(',')]
for rank, sample in pairs:
    ctx.set_option("join_rank", rank); ctx.set_option("join_sample", sample)
    best = None
    for rep in range(3):
        ctx.profile(True)
        torch.cuda.synchronize(); t0 = time.time()
        A.knn_cosine(Xd, gp, ctx=ctx)
        torch.cuda.synchronize(); t1 = time.time()
        prof = ctx.profile_all()
        if best is None or t1 - t0 < best[0]: best = (t1 - t0, prof)
    p = {k: round(v[0], 2) for k, v in best[1].items()}
    print("rank %d sample %d: %.1f ms  cand/row %.1f uncertified %d 1a %d | %s" % (rank, sample, best[0] * 1e3, ctx.stat("knn_join_candidates") / n,
          ctx.stat("knn_uncertified_rows"), ctx.stat("knn_uncertified_rows_stage1a"), p), "stream retries", ctx.stat("stream_scan_retries"), "overflow", ctx.stat("stream_scan_overflow_rows"), flush=True)
