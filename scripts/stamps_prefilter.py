"""Cycle stamps of the int8 first pass (debug bit 8; printed by the library on stderr) and its time with the DMA of the next
stage switched off (debug 512) or the test switched off (debug 256): usage stamps_prefilter.py [n]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
os.environ["ARROWSPACE_ALLOW_ABLATION"] = "1"
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
Xd, cent = synth.mixture_device(n, 128, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
A.build_laplacian_matrix(Xd, gp, ctx=ctx); ctx.synchronize()
for dbg in (0, 8, 0, 8):
    ctx.set_option("debug", dbg)
    ctx.profile(True)
    try:
        A.build_laplacian_matrix(Xd, gp, ctx=ctx)
    except Exception as e:  # noqa: BLE001 -- ablation bits stop the build behind the join
        print("   (", str(e)[:80], ")")
    ctx.synchronize()
    prof = ctx.profile_all(); ctx.profile(False)
    print("debug=%d: first pass %.3f ms" % (dbg, prof["knn_join_prefilter"][0]), flush=True)
ctx.set_option("debug", 0)
