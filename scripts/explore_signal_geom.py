"""Item-graph lambda: geometry sweep of the graph pass (chunk rows x waves) on the by-cluster order."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
from arrowspace_amd.api import compute_taumode_signals
ctx = A.default_context()
n = 1000000
X, cent = synth.mixture(n, 128, seed=42)
Xd = torch.from_numpy(X).cuda()
gl = A.build_laplacian_matrix(Xd, A.GraphParams(0.5, 16, 16, 2.0, 0.25), ctx=ctx)
order = np.argsort(np.arange(n) % 256, kind="stable").astype(np.uint32)
T = A.TauMode
for ch, waves in ((8, 125000), (16, 62504), (8, 62504), (4, 250000), (32, 31256), (4, 0), (4, 2048), (4, 1024), (16, 2048), (64, 2048), (1, 2048)):
    ctx.set_option("signal_chunk_rows", ch); ctx.set_option("signal_waves", waves)
    gl.matrix.set_order(order)
    for it in range(4):
        if it == 1: ctx.profile(True)
        compute_taumode_signals(Xd, gl.matrix, T.Fixed(0.5), ctx=ctx); ctx.synchronize()
    prof = ctx.profile_all(); ctx.profile(False)
    print("chunk=%d waves=%d (used %d): sums %.3f ms reduce %.3f ms" % (ch, waves, ctx.stat("signal_stream_waves"), prof["signal_sums"][0] / prof["signal_sums"][1], prof["signal_reduce"][0] / prof["signal_reduce"][1]), flush=True)
