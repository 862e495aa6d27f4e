"""Randomised comparison of the clustering pre-stage (csrc/prestage.hip, csrc/centroids.hip) with oracle/prestage.c: incremental
clustering (the chain kernel, the batch rule, the matrix-pipe nearest-centroid search), nearest_centroids through the candidate path,
Two-NN ratios on the matrix pipe.  Everything bit for bit.  usage: python scripts/stress_prestage.py [cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np
import arrowspace_amd as A
from oracle import oracle_ctypes as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = A.default_context()
bad = 0
for case in range(cases):
    kind = case % 3
    f = int(rng.choice([2, 3, 8, 17, 32, 64, 100, 128, 130, 200, 256]))
    if kind == 0:      # incremental clustering
        n = int(rng.integers(200, 60000))
        k = int(rng.choice([2, 5, 30, 100, 317, 1200]))
        while n * k * f > 3e9: n //= 2
        batch = int(rng.choice([1, 64, 1000, 4096]))
        nb = max(2, min(k, int(rng.integers(2, 80))))
        many = case % 2 == 0   # every other clustering case: many clusters and large batches (the matrix-pipe nearest-centroid search)
        if many:
            k = int(rng.choice([80, 150, 317])); batch = int(rng.choice([2048, 4096])); nb = k + 20
            n = max(n, 12 * batch)
            while n * k * f > 6e9: f = max(2, f // 2)
        C0 = rng.normal(size=(nb, f)) * 3
        X = C0[rng.integers(0, nb, size=n)] + float(rng.choice([0.2, 0.5, 1.0])) * rng.normal(size=(n, f))
        if rng.random() < 0.5: X[n // 2] = X[n // 3]
        if rng.random() < 0.3: X[-5:] += 1e3
        if rng.random() < 0.2: X[n // 7, f // 2] = np.nan
        d2 = np.sum((X[:100] - X[100:200]) ** 2, axis=1)
        radius = (float(rng.choice([0.01, 0.03])) if many else float(rng.choice([0.02, 0.1, 0.4, 1.0]))) * float(np.nanmedian(d2))
        t0 = time.time()
        c, a, s = A.incremental_clustering(X, k, radius, batch, ctx=ctx)
        t1 = time.time()
        oc, oa, osz = O.incremental_clustering(X, k, radius, batch)
        ok = len(c) == len(oc) and np.array_equal(a, oa) and np.array_equal(s, osz) and np.array_equal(c, oc, equal_nan=True)
        print("clustering n=%d f=%d k=%d batch=%d -> %d clusters, %d dropped, seq %d par %d mfma %d (%.2fs): %s" % (
            n, f, k, batch, len(c), int((a < 0).sum()), ctx.stat("clustering_sequential_batches"), ctx.stat("clustering_device_batches"),
            ctx.stat("nearest_mfma_calls"), t1 - t0, "ok" if ok else "MISMATCH"), flush=True)
    elif kind == 1:    # nearest centroids through the candidate path
        n = int(rng.integers(1024, 20000)); nc = int(rng.integers(64, 3000))
        while n * nc * f > 3e9: n //= 2
        n = max(n, 1024)
        Cn = rng.normal(size=(nc, f)) * float(rng.choice([0.1, 1.0, 30.0]))
        if rng.random() < 0.5: Cn = np.tile(Cn[: max(1, nc // 8)], (9, 1))[:nc] + 1e-9 * rng.normal(size=(nc, f))
        X = Cn[rng.integers(0, nc, size=n)] + float(rng.choice([1e-6, 0.1, 1.0])) * rng.normal(size=(n, f))
        if rng.random() < 0.3: X += 1e3; Cn = Cn + 1e3
        X[3] = Cn[nc // 2]
        ctx.set_option("nearest_mfma", 2)
        idx, d2 = A.nearest_centroids(X, Cn, ctx=ctx)
        ctx.set_option("nearest_mfma", 1)
        oi, od = O.nearest_centroids(X, Cn)
        ok = np.array_equal(idx, oi) and np.array_equal(d2, od, equal_nan=True)
        print("nearest n=%d f=%d nc=%d: %s" % (n, f, nc, "ok" if ok else "MISMATCH"), flush=True)
    else:              # Two-NN on the matrix pipe
        n = int(rng.integers(65536, 120000)); ns = int(rng.choice([60, 500]))
        while n * ns * f > 3e9: ns //= 2
        nb = int(rng.integers(2, 60))
        C0 = rng.normal(size=(nb, f)) * 3
        X = C0[rng.integers(0, nb, size=n)] + float(rng.choice([0.05, 0.5])) * rng.normal(size=(n, f))
        ids = O.pre_sample_indices(n, ns, int(rng.integers(1, 1000)))
        if rng.random() < 0.5: X[(int(ids[1]) + 9) % n] = X[int(ids[1])]
        if rng.random() < 0.3: X[int(ids[2])] += 300.0
        got = A.two_nn_ratios(X, ids, ctx=ctx)
        want = O.two_nn_ratios(X, ids)
        ok = np.array_equal(got, want, equal_nan=True)
        print("two_nn n=%d f=%d ns=%d mfma %d redo %d: %s" % (n, f, ns, ctx.stat("two_nn_mfma"), ctx.stat("two_nn_mfma_overflow"), "ok" if ok else "MISMATCH"), flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
