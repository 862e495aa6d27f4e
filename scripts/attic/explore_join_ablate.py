import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = 1000000
X, cent = synth.mixture(n, 128, seed=42)
Xd = torch.from_numpy(X).cuda()
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
for dbg in (0, 1, 2, 0):
    ctx.set_option("debug", dbg)
    ctx.profile(True)
    try:
        A.knn_cosine(Xd, gp, ctx=ctx)
    except Exception as e:
        print("err", e)
    torch.cuda.synchronize()
    prof = ctx.profile_all()
    print("debug=%d" % dbg, {k: round(v[0] / v[1], 1) for k, v in prof.items() if "join" in k})
