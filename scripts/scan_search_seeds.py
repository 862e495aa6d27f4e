"""A query batch against indices that differ in their lambdas only (seeded draws): wall time of the first and the second search, queries
left open.  Found with it (round 5): one draw in 15 left 2 - 8 of 100 000 queries open at 1 M x 768, and the one-wave-per-query exact
scan they took made a 41 ms batch take 1.8 s.  usage: scan_search_seeds.py n d nq [seeds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
n, d, nq = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = A.default_context()
X, cent = synth.mixture_device(n, d, "cuda", seed=42)
qidx = synth.query_indices(n, nq, seed=42)
qi = torch.from_numpy(qidx.astype(np.int64)).cuda()
Q = X[qi] * 1.02
for seed in range(int(sys.argv[4]) if len(sys.argv) > 4 else 24):
    torch.manual_seed(seed)
    lam = torch.rand(n, dtype=torch.float64, device="cuda") * 0.4 + 0.01
    lq = lam[qi] + 0.01
    asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
    asp.update_lambdas(lam)
    ts = []
    for call in range(2):
        t0 = time.perf_counter()
        asp.search_lambda_aware_batch(Q, lq, 10, 0.7)
        ctx.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("seed %d: %.1f / %.1f ms; uncertified %d, exact scan rows %d, listed %d, candidates %d, overflow %d" % (
        seed, ts[0], ts[1], ctx.stat("search_uncertified_rows"), ctx.stat("search_exact_rows"), ctx.stat("search_scan_f16_pairs_listed"), ctx.stat("search_scan_candidates"),
        ctx.stat("search_scan_overflow_rows")), flush=True)
    del asp
