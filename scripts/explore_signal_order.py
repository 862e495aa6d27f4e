"""Item-graph lambda: the join's own row order as attached by as_build_laplacian, with and without the chaining of its
blocks by the graph (signal_order_refine); the preparation of the record stream is timed too.  usage: [n]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
from arrowspace_amd.api import compute_taumode_signals
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
Xd, cent = synth.mixture_device(n, 128, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
ref = None
for refine in (1, 0, 1, 0):
    ctx.set_option("signal_order_refine", refine)
    gl = A.build_laplacian_matrix(Xd, gp, ctx=ctx)
    ctx.synchronize()
    ctx.profile(True)
    lam = compute_taumode_signals(Xd, gl.matrix, A.TauMode.Fixed(0.5), ctx=ctx); ctx.synchronize()
    prep = ctx.profile_all().get("signal_stream_prepare", (0, 1))
    ctx.profile(False)
    for it in range(4):
        if it == 1: ctx.profile(True)
        lam = compute_taumode_signals(Xd, gl.matrix, A.TauMode.Fixed(0.5), ctx=ctx); ctx.synchronize()
    prof = ctx.profile_all(); ctx.profile(False)
    lam = lam.cpu().numpy()
    if ref is None: ref = lam
    print("refine=%d (used %d): sums %.3f ms, stream preparation %.2f ms, max rel diff to the first %.2e" % (
        refine, ctx.stat("signal_stream_order_refined"), prof["signal_sums"][0] / prof["signal_sums"][1], prep[0] / max(prep[1], 1),
        float(np.max(np.abs(lam - ref) / np.abs(ref)))), flush=True)
