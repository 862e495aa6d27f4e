"""Timing experiment: the scoring kernel with a shortened chain (debug 64: outputs wrong on purpose)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = 1000000
Xd, cent = synth.mixture_device(n, 128, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
for dbg in (0, 64 + 256 * 1, 64 + 256 * 2, 64 + 256 * 3, 64 + 256 * 7):
    ctx.set_option("debug", dbg)
    for rep in range(2):
        A.knn_cosine(Xd, gp, ctx=ctx); torch.cuda.synchronize()
