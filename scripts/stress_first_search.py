import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np
import arrowspace_amd as A
from oracle import oracle_ctypes as O
P = np.array(json.load(open(os.path.join(ROOT, "tests/golden/proteins_64x24.json")))["rows"])
ctx = A.default_context()
ctx.set_option("knn_mfma", 0); ctx.set_option("search_mfma", 0)
lam = np.linspace(0.01, 0.9, 64)
asp = A.ArrowSpace(P, A.TauMode.Median, ctx)
asp.update_lambdas(lam)
q = P[3] * 1.02
res = asp.search_lambda_aware(A.ArrowItem(q, 0.37), 3, 0.7)
oi, osc = O.search_lambda_aware(P, lam, q, 0.37, 3, 0.7)
ok = [r[0] for r in res] == oi.tolist() and np.array_equal([r[1] for r in res], osc)
if not ok:
    idx, sc = asp.search_lambda_aware_batch(q.reshape(1, -1), np.array([0.37]), 64, 0.0)
    lam_used = np.empty(64); lam_used[idx[0].astype(int)] = sc[0]
    print("MISMATCH", res, oi, osc)
    print("score(alpha=0) by item:", np.round(lam_used, 3).tolist())
    print("expected:", np.round(1 - np.minimum(np.abs(0.37 - lam), 1), 3).tolist())
print("ok" if ok else "FAIL")
