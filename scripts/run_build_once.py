"""Three item-graph builds + lambdas at n x 128 (for rocprofv3 --kernel-trace: scripts/trace_gaps.py reads the last one).
usage: run_build_once.py [n]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
Xd, cent = synth.mixture_device(n, 128, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
for it in range(3):
    gl = A.build_laplacian_matrix(Xd, gp, ctx=ctx)
    ctx.synchronize()
    del gl
print("done")
