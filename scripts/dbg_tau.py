import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np
import arrowspace_amd as A
from oracle import oracle_ctypes as O
ctx = A.default_context()
n_len = 200000
rng = np.random.default_rng(17)
X = rng.standard_normal((n_len, 10))
X[:, 1] = np.exp(4.0 * X[:, 1])
X[:, 2] = rng.standard_cauchy(n_len)
X[:, 3] = np.abs(X[:, 3]) * 1e-300
X[::7, 4] = np.nan
X[::5, 5] = np.inf
X[::3, 5] = -np.inf
X[:, 6] = -np.abs(X[:, 6])
X[:, 7] = np.arange(n_len)[::-1] * 0.5
X[:, 8] = np.round(X[:, 8], 4)
for c in range(10):
    for tm, om, oa in ((A.TauMode.Median, O.TAU_MEDIAN, 0.0), (A.TauMode.Percentile(0.95), O.TAU_PERCENTILE, 0.95), (A.TauMode.Percentile(1.0), O.TAU_PERCENTILE, 1.0), (A.TauMode.Percentile(0.0), O.TAU_PERCENTILE, 0.0)):
        got = A.select_tau(np.ascontiguousarray(X[:, c:c + 1].T), tm, ctx=ctx)
        print(c, tm.kind, tm.arg, "fallback", ctx.stat("signal_tau_bracket_fallback_columns"), got[0] == O.select_tau(X[:, c], om, oa))
