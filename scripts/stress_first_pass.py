"""Randomised stress of the int8 first pass (join and search): tables / results with it must equal those without, over random
sizes, widths, cluster structure, duplicates and zero rows.  usage: stress_first_pass.py [cases] [seed]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx.set_option("join_min_rows", 512)
bad = 0
for c in range(cases):
    big = rng.random() < 0.25
    n = int(rng.integers(66000, 220000)) if big else int(rng.integers(600, 9000))
    d = int(rng.choice([8, 24, 31, 32, 33, 48, 64, 65, 96, 100, 127, 128, 129, 200, 256, 384, 500, 768, 1000, 1024]))
    if big and d > 256: d = int(rng.choice([64, 128, 192]))
    topk = int(rng.integers(3, 40)); eps = float(rng.choice([0.3, 0.5, 0.9]))
    kind = rng.integers(0, 4)
    if kind == 0:
        X = rng.standard_normal((n, d))
    else:
        X, _ = synth.mixture(n, d, seed=int(rng.integers(1, 1 << 30)), n_clusters=int(rng.integers(2, 80)), rank=int(rng.integers(1, 9)))
    X[rng.integers(0, n, 5)] = X[rng.integers(0, n)]
    X[rng.integers(0, n)] = 0.0
    if rng.random() < 0.3: X *= rng.uniform(1e-3, 1e3, size=(n, 1))
    gp = A.GraphParams(eps, 16, topk, 2.0, 0.25)
    mode = int(rng.choice([1, 2, 3]))
    if os.environ.get("ONLY") and int(os.environ["ONLY"]) != c: continue
    ctx.set_option("join_prefilter", 0)
    i0, d0 = A.knn_cosine(X, gp)
    ctx.set_option("join_prefilter", mode); ctx.set_option("join_prefilter_min_rows", 0)
    for kv in os.environ.get("OPTS", "").split(","):
        if "=" in kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    for rep in range(int(os.environ.get("REPEAT", "1"))):
        i1, d1 = A.knn_cosine(X, gp)
        if os.environ.get("CHECKSUM"): print("   rep %d theta checksum %d leftover %d uncertified %d cand %d" % (rep, ctx.stat("knn_join_theta_checksum"), ctx.stat("knn_uncertified_rows_stage1a"), ctx.stat("knn_uncertified_rows"), ctx.stat("knn_join_candidates")), flush=True)
        if not (np.array_equal(i0, i1) and np.array_equal(d0, d1)):
            bad_rows = np.nonzero((i0 != i1).any(axis=1) | (d0 != d1).any(axis=1))[0]
            print("   rep %d: %d rows differ, first %s; stats: leftover %d stream rows %d filter %d items %d overflow %d grouped %d" % (
                rep, len(bad_rows), bad_rows[:5], ctx.stat("knn_uncertified_rows_stage1a"), ctx.stat("knn_stream_scan_rows"), ctx.stat("stream_scan_filter"),
                ctx.stat("stream_scan_filter_items"), ctx.stat("stream_scan_filter_overflow"), ctx.stat("knn_join_theta_grouped")), flush=True)
            r = int(bad_rows[0])
            print("   row %d: norm %.6g, rows equal to it: %d; last base d %r got d %r; base idx tail %r got %r" % (
                r, float(np.linalg.norm(X[r])), int((X == X[r]).all(axis=1).sum()), d0[r][-3:].tolist(), d1[r][-3:].tolist(), i0[r][-3:].tolist(), i1[r][-3:].tolist()), flush=True)
            print("   base idx", i0[r][:12], "...", i0[r][-6:], "\n   got  idx", i1[r][:12], "...", i1[r][-6:], "\n   base d", d0[r][-4:], "got d", d1[r][-4:],
                  "first diff at", int(np.nonzero((i0[r] != i1[r]) | (d0[r] != d1[r]))[0][0]), flush=True)
            break
    used, rot = ctx.stat("knn_join_prefilter"), ctx.stat("knn_join_prefilter_rotated")
    ctx.set_option("join_prefilter", 1); ctx.set_option("join_prefilter_min_rows", 65536)
    ok = np.array_equal(i0, i1) and np.array_equal(d0, d1)
    bad += 0 if ok else 1
    print("join #%d n=%d d=%d topk=%d eps=%.1f kind=%d mode=%d used=%d rotated=%d join=%d: %s" % (c, n, d, topk, eps, kind, mode, used, rot, ctx.stat("knn_join"), "ok" if ok else "MISMATCH"), flush=True)
for c in range(max(2, cases // 6)):
    n = int(rng.integers(66000, 180000)); d = int(rng.choice([33, 64, 128, 128, 128])); nq = int(rng.integers(40, 6000)); k = int(rng.integers(1, 30))
    alpha = float(rng.choice([0.2, 0.7, 1.0]))
    X, _ = synth.mixture(n, d, seed=int(rng.integers(1, 1 << 30)), n_clusters=int(rng.integers(2, 80)), rank=int(rng.integers(1, 9)))
    lam = rng.random(n); qi = rng.integers(0, n, nq); Q = X[qi] * 1.01 + rng.normal(size=(nq, d)) * 0.01; lq = lam[qi] + rng.normal(size=nq) * 0.03
    res = []
    for order, pre in ((0, 0), (2, 1)):
        ctx.set_option("search_scan_order", order); ctx.set_option("search_prefilter", pre)
        asp = A.ArrowSpace(X, A.TauMode.Median, ctx); asp.update_lambdas(lam)
        res.append(asp.search_lambda_aware_batch(Q, lq, k, alpha))
        st = (ctx.stat("search_scan"), ctx.stat("search_scan_order"), ctx.stat("search_scan_prefilter"))
    ctx.set_option("search_scan_order", 1); ctx.set_option("search_prefilter", 1)
    ok = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    if d == 128 and k <= 64:  # small batches on the index searched before: the int8 filter of the streaming scan
        for b in (1, int(rng.integers(2, 33))):
            o = int(rng.integers(0, max(1, nq - b)))
            sm = asp.search_lambda_aware_batch(Q[o:o + b], lq[o:o + b], k, alpha)
            bb = min(b, nq - o)
            ok = ok and np.array_equal(res[0][0][o:o + bb], sm[0]) and np.array_equal(res[0][1][o:o + bb], sm[1])
            st = st + (ctx.stat("search_stream_scan"), ctx.stat("stream_scan_filter"), ctx.stat("stream_scan_filter_items"))
    bad += 0 if ok else 1
    print("search n=%d d=%d nq=%d k=%d alpha=%.1f scan/order/prefilter=%s: %s" % (n, d, nq, k, alpha, st, "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
