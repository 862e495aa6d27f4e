"""Randomised stress of the row order that draws further pivots (round 5) and of the whole-sample thresholds of the rows it cannot
place: the neighbour table with them must equal the table without ("join_order_cover" 0, "join_theta_far" 0) and the table of the
join on the rows as given ("join_row_order" 0), over random sizes, widths, many clusters (more than pivots), ranks, depths,
duplicates, zero rows and row scales.  usage: stress_row_order.py [cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    n = int(rng.choice([rng.integers(66000, 140000), rng.integers(140000, 420000), rng.integers(245760, 262144 + 600)]))
    d = int(rng.choice([24, 64, 100, 128, 128, 200, 384]))
    if n > 200000 and d > 128:
        d = 128
    clusters = int(rng.choice([rng.integers(40, 300), rng.integers(300, 1500), rng.integers(1500, 7000)]))
    rank = int(rng.choice([4, 8, 16, 32, 64]))
    topk = int(rng.integers(3, 33))
    X, _ = synth.mixture_device(n, d, "cuda", seed=int(rng.integers(1, 1 << 30)), n_clusters=clusters, rank=min(rank, d))
    X[torch.from_numpy(rng.integers(0, n, 5)).cuda()] = X[int(rng.integers(0, n))].clone()
    X[int(rng.integers(0, n))] = 0.0
    if rng.random() < 0.3:
        X *= torch.from_numpy(rng.uniform(1e-3, 1e3, size=(n, 1))).cuda()
    gp = A.GraphParams(float(rng.choice([0.3, 0.5, 0.9])), 16, topk, 2.0, 0.25)
    pivots = int(rng.choice([1024, 1024, 256, 4096]))
    tabs = []
    for name, opts in (("plain", {"join_row_order": 0}), ("fixed", {"join_order_cover": 0, "join_theta_far": 0, "join_order_pivots": pivots}),
                       ("grown", {"join_order_pivots": pivots})):
        for k, v in (("join_row_order", 1), ("join_order_cover", 150), ("join_theta_far", 1), ("join_order_pivots", 1024)):
            ctx.set_option(k, v)
        for k, v in opts.items():
            ctx.set_option(k, v)
        t0 = time.perf_counter()
        i1, d1 = A.knn_cosine(X, gp)
        tabs.append((name, i1, d1, (time.perf_counter() - t0) * 1e3, ctx.stat("knn_join"), ctx.stat("knn_join_order_pivots"), ctx.stat("knn_join_order_cover_rounds"),
                     ctx.stat("knn_join_theta_far_rows"), ctx.stat("knn_join_theta_dip_rows"), ctx.stat("knn_join_theta_grouped")))
    ok = all(np.array_equal(tabs[0][1], t[1]) and np.array_equal(tabs[0][2], t[2]) for t in tabs[1:])
    bad += 0 if ok else 1
    print("#%d n=%d d=%d clusters=%d rank=%d topk=%d pivots=%d: %s  " % (c, n, d, clusters, rank, topk, pivots, "ok" if ok else "MISMATCH"),
          [(t[0], "%.0f ms" % t[3], "join %d pivots %d rounds %d far %d dip %d grouped %d" % t[4:]) for t in tabs], flush=True)
    del X
for k, v in (("join_row_order", 1), ("join_order_cover", 150), ("join_theta_far", 1), ("join_order_pivots", 1024)):
    ctx.set_option(k, v)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
