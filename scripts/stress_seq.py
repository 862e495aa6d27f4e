import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np
import arrowspace_amd as A
flags = sys.argv[1] if len(sys.argv) > 1 else "pqs"   # p: prepare query, s: search, v: lambda_v2 on
g = json.load(open(os.path.join(ROOT, "tests/golden/oracle_vectors.json")))
unhex = lambda a: np.array([float.fromhex(x) for x in a])
ds = {k: np.array([[float.fromhex(x) for x in r] for r in v]) for k, v in g["datasets"].items()}
ctx = A.default_context()
ctx.set_option("knn_mfma", 0); ctx.set_option("search_mfma", 0)
ctx.set_option("lambda_v2", 1 if "v" in flags else 0)
bad = []
for c in g["cases"]:
    if "lambdas" not in c: continue
    rows = ds[c["dataset"]]
    nodes = np.ascontiguousarray(rows.T) if c["transpose"] else rows
    csr = (np.array(c["csr"]["indptr"], dtype=np.uint64), np.array(c["csr"]["indices"], dtype=np.uint64), unhex(c["csr"]["data"]))
    L = A.CsrMatrix.from_host(ctx, *csr)
    gl = A.GraphLaplacian(nodes, L, len(rows), None)
    asp = A.ArrowSpace(rows, A.TauMode(c["tau_mode"], c["tau_arg"]), ctx)
    asp.compute_taumode(gl)
    want = unhex(c["lambdas"])
    if not np.allclose(asp.lambdas(), want, rtol=1e-12, atol=1e-300):
        bad.append(("lam", c["name"]))
    for q in c.get("queries", []):
        if "p" in flags:
            try:
                lq = asp.prepare_query_item(q["q"], gl)
                if abs(lq - float.fromhex(q["lambda_q"])) > 1e-12 * abs(lq): bad.append(("lq", c["name"]))
            except Exception as e:
                bad.append(("lq-exc", c["name"]))
        if "s" in flags and "idx" in q:
            asp.update_lambdas(want)
            try:
                res = asp.search_lambda_aware(A.ArrowItem(q["q"], float.fromhex(q["lambda_q"])), q["k"], q["alpha"])
                if [r[0] for r in res] != q["idx"]:
                    bad.append(("search", c["name"], [r[0] for r in res]))
                    from oracle import oracle_ctypes as O
                    lq = float.fromhex(q["lambda_q"]); qq = np.array(q["q"])
                    res2 = asp.search_lambda_aware(A.ArrowItem(q["q"], lq), q["k"], q["alpha"])
                    print("  retry same index:", [r[0] for r in res2])
                    idx, sc = asp.search_lambda_aware_batch(qq.reshape(1, -1), np.array([lq]), len(rows), q["alpha"])
                    got = np.empty(len(rows)); got[idx[0].astype(int)] = sc[0]
                    exp = np.array([O.lambda_similarity(qq, lq, rows[i], want[i], q["alpha"]) for i in range(len(rows))])
                    print("  full-k retry max |diff| vs oracle:", np.abs(got - exp).max())
                    print("  first result scores:", [r[1] for r in res], "expected top:", sorted(exp)[-3:][::-1])
                    wrong = [r[0] for r in res]
                    print("  oracle scores of returned ids:", [exp[i] for i in wrong])
            except Exception as e:
                bad.append(("search-exc", c["name"]))
print(flags, "BAD" if bad else "ok", bad[:4])
