import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np
import arrowspace_amd as A
opt = int(sys.argv[1])
if len(sys.argv) > 2:
    import torch
    print('torch cuda', torch.cuda.is_available())
g = json.load(open(os.path.join(ROOT, "tests/golden/oracle_vectors.json")))
unhex = lambda a: np.array([float.fromhex(x) for x in a])
ds = {k: np.array([[float.fromhex(x) for x in r] for r in v]) for k, v in g["datasets"].items()}
ctx = A.default_context()
ctx.set_option("lambda_v2", opt)
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
        bad.append((c["name"], int((~np.isclose(asp.lambdas(), want, rtol=1e-12, atol=1e-300)).sum())))
print("opt", opt, "BAD" if bad else "ok", bad)
