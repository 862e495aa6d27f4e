import sys, os
ROOT = "/root/repo"
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
gl.matrix.set_order(np.argsort(np.arange(n) % 256, kind="stable").astype(np.uint32))
T = A.TauMode
for S in (128, 64, 32, 16):
    Xs = Xd[:, :S].contiguous()
    for it in range(4):
        if it == 1: ctx.profile(True)
        compute_taumode_signals(Xs, gl.matrix, T.Fixed(0.5), ctx=ctx); ctx.synchronize()
    prof = ctx.profile_all(); ctx.profile(False)
    ms, cnt = prof["signal_sums"]
    print("S=%d sums %.3f ms  -> gather rate %.2f TB/s" % (S, ms / cnt, 8.5e6 * S * 8 / (ms / cnt) / 1e9))
