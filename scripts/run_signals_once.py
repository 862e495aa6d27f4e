"""One item-graph lambda call (for rocprofv3): usage run_signals_once.py [n] [order: cluster|natural|random] [iters]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
from arrowspace_amd.api import compute_taumode_signals
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
order = sys.argv[2] if len(sys.argv) > 2 else "cluster"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
X, cent = synth.mixture(n, 128, seed=42)
Xd = torch.from_numpy(X).cuda()
gl = A.build_laplacian_matrix(Xd, A.GraphParams(0.5, 16, 16, 2.0, 0.25), ctx=ctx)
if order == "cluster":
    gl.matrix.set_order(np.argsort(np.arange(n) % 256, kind="stable").astype(np.uint32))
elif order == "random":
    gl.matrix.set_order(np.random.default_rng(1).permutation(n).astype(np.uint32))
for it in range(iters):
    lam = compute_taumode_signals(Xd, gl.matrix, A.TauMode.Fixed(0.5), ctx=ctx)
    ctx.synchronize()
print("nnz", gl.matrix.nnz(), "lambda[0..3]", lam[:4].cpu().numpy())
