"""Randomised stress of the order statistics of long columns (Median / Percentile tau from a sampled bracket and one pass, round 5):
as_select_tau on random columns -- smooth of any scale, heavy tails, ties of every density, non-finite entries, sorted and
periodic data, even and odd counts -- must return the oracle's select_tau bit for bit, with the bracket path and with the digit
passes.  usage: stress_tau.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np
import arrowspace_amd as A
from oracle import oracle_ctypes as O
ctx = A.default_context()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
T = A.TauMode
bad = 0
for c in range(cases):
    n = int(rng.choice([rng.integers(131072, 140000), rng.integers(140000, 700000), 262144, 262145]))
    S = int(rng.integers(1, 9))
    X = np.empty((n, S))
    kinds = []
    for s in range(S):
        k = int(rng.integers(0, 9))
        kinds.append(k)
        if k == 0: col = rng.standard_normal(n) * 10.0 ** rng.integers(-200, 200)
        elif k == 1: col = rng.standard_cauchy(n)
        elif k == 2: col = np.round(rng.standard_normal(n), int(rng.integers(0, 6)))              # ties of every density
        elif k == 3: col = rng.integers(0, int(rng.integers(2, 2000)), size=n).astype(np.float64)  # few distinct values
        elif k == 4: col = np.sort(rng.standard_normal(n))[:: int(rng.choice([1, -1]))]           # sorted either way
        elif k == 5: col = np.sin(np.arange(n) * (2 * np.pi / int(rng.choice([64, 61, 122, 8192, 1000]))))  # periodic in the row index
        elif k == 6: col = np.exp(rng.standard_normal(n) * 20.0)
        elif k == 7: col = -np.abs(rng.standard_normal(n))
        else: col = np.full(n, float(rng.standard_normal()))
        col = col.astype(np.float64)
        r = rng.random()
        if r < 0.25: col[rng.random(n) < rng.random() * 0.5] = np.nan
        elif r < 0.4: col[:: int(rng.integers(2, 9))] = np.inf * rng.choice([1.0, -1.0])
        elif r < 0.45: col[:] = np.nan
        X[:, s] = col
    Xt = np.ascontiguousarray(X.T)
    pp = float(rng.choice([0.0, 0.01, 0.5, 0.95, 0.999, 1.0, rng.random()]))
    for tm, om, oa in ((T.Median, O.TAU_MEDIAN, 0.0), (T.Percentile(pp), O.TAU_PERCENTILE, pp)):
        want = np.array([O.select_tau(X[:, s], om, oa) for s in range(S)])
        ctx.set_option("signal_tau_bracket", 1)
        got = A.select_tau(Xt, tm, ctx=ctx)
        fb = ctx.stat("signal_tau_bracket_fallback_columns")
        ctx.set_option("signal_tau_bracket", 0)
        got0 = A.select_tau(Xt, tm, ctx=ctx)
        ctx.set_option("signal_tau_bracket", 1)
        ok = np.array_equal(got, want) and np.array_equal(got0, want)
        bad += 0 if ok else 1
        print("#%d n=%d kinds=%s mode=%s(%.3f) fallback columns %d: %s" % (c, n, kinds, "median" if om == O.TAU_MEDIAN else "percentile", oa, fb, "ok" if ok else "MISMATCH %s %s %s" % (got, got0, want)), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
