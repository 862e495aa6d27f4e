import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/arrowspace-rs_amd")
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
from oracle import oracle_ctypes as O
ctx = A.default_context()
n, d = 200_000, 128
X, _ = synth.mixture(n, d, seed=3, n_clusters=64, rank=8)
X[100] = X[7]; X[50000] = X[7]; X[9] = 0.0
lam = np.random.default_rng(1).random(n)
bad = 0
for nq in (1, 2, 3, 4, 5, 8, 9, 32):
    qi = np.random.default_rng(nq).integers(0, n, nq)
    Q = X[qi] * 1.02; lq = lam[qi] + 0.01
    res = {}
    for quick in (0, 1):
        ctx.set_option("search_stream_quick", quick)
        asp = A.ArrowSpace(X, A.TauMode.Median, ctx); asp.update_lambdas(lam)
        res[quick] = asp.search_lambda_aware_batch(Q, lq, 10, 0.7)
        st = (ctx.stat("search_stream_scan"), ctx.stat("stream_scan_quick"), ctx.stat("stream_scan_filter_items"))
        print("nq", nq, "quick", quick, "stats", st)
    ok = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    for j in range(min(nq, 3)):
        oi, osc = O.search_lambda_aware(X, lam, Q[j], lq[j], 10, 0.7)
        ok = ok and np.array_equal(res[1][0][j].astype(np.int64), oi.astype(np.int64)) and np.array_equal(res[1][1][j], osc)
    print("nq", nq, "OK" if ok else "MISMATCH"); bad += 0 if ok else 1
ctx.set_option("search_stream_quick", 1)
sys.exit(bad)
