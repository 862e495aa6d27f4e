"""Dev timing of the search at other widths than 128 (configs[4]: 100 k queries x 1 M x 768).  usage: explore_search_wide.py [n] [d] [nq]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
Xd, cent = synth.mixture_device(n, d, "cuda", seed=42)
g = torch.Generator(device="cuda"); g.manual_seed(7)
lam = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 0.3 + 0.05
qi = torch.randint(0, n, (nq,), device="cuda", generator=g)
Q = (Xd[qi] * 1.01).contiguous()
lq = (lam[qi] * 1.02).contiguous()
asp = A.ArrowSpace(Xd, A.TauMode.Median, ctx=ctx)
asp.update_lambdas(lam)
for rep in range(3):
    ctx.profile(True)
    torch.cuda.synchronize(); t0 = time.time()
    idx, sc = asp.search_lambda_aware_batch(Q, lq, 10, 0.7)
    torch.cuda.synchronize(); t1 = time.time()
    prof = ctx.profile_all()
    print("n=%d d=%d nq=%d: %.1f ms -> %.0f QPS" % (n, d, nq, (t1 - t0) * 1e3, nq / (t1 - t0)), {k: round(v[0], 2) for k, v in prof.items()}, flush=True)
