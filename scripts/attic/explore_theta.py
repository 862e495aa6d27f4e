import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth, distributed as D
ctx = A.default_context()
ctx.set_option("join_min_rows", 512)
n, d = 20000, 128
X, _ = synth.mixture(n, d, seed=42)
Xd = torch.from_numpy(X).cuda()
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
ops = D.HipOps(ctx)
th = {}
for bf in (0, 1):
    ctx.set_option("join_bf16", bf)
    th[bf] = ops.join_thresholds(Xd, 0, n, gp).cpu().numpy()
print("f32  theta:", th[0][:8], np.percentile(th[0], [1, 50, 99]))
print("bf16 theta:", th[1][:8], np.percentile(th[1], [1, 50, 99]))
# exact check for row 0: products with the sample
U = X / np.linalg.norm(X, axis=1, keepdims=True)
s = U @ U[0]
print("row0 best cosines overall:", np.sort(s)[-8:])
