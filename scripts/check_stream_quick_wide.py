"""Small-batch search and leftover graph rows of WIDE items through the quick pass (stream_quick_wide_kernel): equality with the
staged path (search_stream_quick = 0) and with the oracle; timing of a one-query search at 1 M x 768."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
from oracle import oracle_ctypes as O
ctx = A.default_context()
bad = 0
for d in (256, 768, 1024):
    n = 90_000
    X, _ = synth.mixture(n, d, seed=5, n_clusters=40, rank=6)
    X[100] = X[7]; X[50000] = X[7]; X[9] = 0.0
    lam = np.random.default_rng(1).random(n)
    for nq in (1, 2, 3, 5, 9):
        qi = np.concatenate([[7], np.random.default_rng(nq).integers(0, n, nq)])[:nq]
        Q = X[qi] * 1.02; lq = lam[qi] + 0.01
        res = {}
        for quick in (0, 1):
            ctx.set_option("search_stream_quick", quick)
            asp = A.ArrowSpace(X, A.TauMode.Median, ctx); asp.update_lambdas(lam)
            res[quick] = asp.search_lambda_aware_batch(Q, lq, 10, 0.7)
            st = (ctx.stat("search_stream_scan"), ctx.stat("stream_scan_quick"))
        ok = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and st == (1, 1)
        for j in range(min(nq, 2)):
            oi, osc = O.search_lambda_aware(X, lam, Q[j], lq[j], 10, 0.7)
            ok = ok and np.array_equal(res[1][0][j].astype(np.int64), oi.astype(np.int64)) and np.array_equal(res[1][1][j], osc)
        print("d", d, "nq", nq, st, "OK" if ok else "MISMATCH", flush=True); bad += 0 if ok else 1
ctx.set_option("search_stream_quick", 1)
if len(sys.argv) > 1:
    n, d = 1_000_000, 768
    Xd, _ = synth.mixture_device(n, d, "cuda", seed=42)
    lam = torch.rand(n, dtype=torch.float64, device="cuda")
    asp = A.ArrowSpace(Xd, A.TauMode.Median, ctx); asp.update_lambdas(lam)
    Q = (Xd[[5]] * 1.02).contiguous(); lq = lam[[5]] + 0.01
    for quick in (1, 0):
        ctx.set_option("search_stream_quick", quick)
        for _ in range(2): asp.search_lambda_aware_batch(Q, lq, 10, 0.7)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): r = asp.search_lambda_aware_batch(Q, lq, 10, 0.7)
        torch.cuda.synchronize(); print("1M x 768 nq=1 quick", quick, "ms per search", (time.perf_counter() - t0) / 5 * 1e3, r[0][0][:3].tolist())
    ctx.set_option("search_stream_quick", 1)
sys.exit(bad)
