"""Kernel-level view of the small-batch search (run under rocprofv3 --kernel-trace --stats).  usage: profile_small_batch.py [nq]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n, d = 1000000, 128
Xd, _ = synth.mixture_device(n, d, "cuda", seed=42)
g = torch.Generator(device="cuda"); g.manual_seed(7)
lam = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 0.3 + 0.05
qi = torch.randint(0, n, (nq,), device="cuda", generator=g)
Q = (Xd[qi] * 1.01).contiguous(); lq = (lam[qi] * 1.02).contiguous()
asp = A.ArrowSpace(Xd, A.TauMode.Median, ctx=ctx); asp.update_lambdas(lam)
for _ in range(200): asp.search_lambda_aware_batch(Q, lq, 10, 0.7)
torch.cuda.synchronize()
