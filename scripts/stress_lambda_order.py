"""Randomised check of the search scan in the order of the lambdas (fresh index, batches of 1024 queries or more): the same
results with `search_lambda_order` 1, 0 and with the lambda-range bound switched off (`debug` bit 22), over random sizes, widths,
alpha, lambda spreads, duplicate lambdas and outlier queries.  usage: stress_lambda_order.py [cases] [seed]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np
import arrowspace_amd as A
from tests import synth
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = A.default_context()
bad = 0
for c in range(cases):
    n = int(rng.integers(33_000, 90_000)); d = int(rng.choice([33, 64, 96, 128])); nq = int(rng.integers(1024, 4000))
    k = int(rng.integers(1, 30)); alpha = float(rng.choice([0.05, 0.3, 0.5, 0.7, 0.95])); spread = float(rng.choice([1e-6, 0.05, 0.4, 2.0, 8.0]))
    X, _ = synth.mixture(n, d, seed=int(rng.integers(1 << 30)), n_clusters=int(rng.integers(4, 60)), rank=int(rng.integers(2, 12)))
    lam = rng.random(n) * spread + 0.01
    lam[::int(rng.integers(3, 11))] = lam[5]
    qi = rng.integers(0, n, size=nq)
    Q = X[qi] * 1.01 + rng.normal(size=(nq, d)) * 0.02
    lq = np.abs(lam[qi] + rng.normal(size=nq) * 0.1 * spread) + 1e-9
    n_out = int(rng.integers(0, 60))
    lq[rng.integers(0, nq, size=n_out)] = lam.max() + float(rng.choice([0.5, 1.5, 10.0]))   # outliers: the lambda term near or at its floor
    res = []
    for order, dbg in ((1, 0), (0, 0), (1, 0x400000)):
        ctx.set_option("search_lambda_order", order); ctx.set_option("debug", dbg)
        asp = A.ArrowSpace(X, A.TauMode.Median, ctx); asp.update_lambdas(lam)
        res.append(asp.search_lambda_aware_batch(Q, lq, k, alpha))
        scan = ctx.stat("search_scan")
    ctx.set_option("debug", 0); ctx.set_option("search_lambda_order", 1)
    same = all(np.array_equal(res[0][0], r[0]) and np.array_equal(res[0][1], r[1]) for r in res[1:])
    bad += 0 if same else 1
    print("n=%d d=%d nq=%d k=%d alpha=%.2f spread=%g outliers=%d scan=%d: %s" % (n, d, nq, k, alpha, spread, n_out, scan, "ok" if same else "MISMATCH"), flush=True)
print("mismatches:", bad)
