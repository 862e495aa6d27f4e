"""Dev timing of the graph pass of the item-graph lambda with a whole slab (two columns per lane, 1-KB gathers) against half
slabs (one column per lane, 512-byte gathers), per row order and rows per wave; the two must agree bit for bit.
usage: explore_signal_half.py [n]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
from arrowspace_amd.api import compute_taumode_signals
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
F = 128
X, cent = synth.mixture(n, F, seed=42)
Xd = torch.from_numpy(X).cuda()
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
gl = A.build_laplacian_matrix(Xd, gp, ctx=ctx)
ctx.synchronize()
nnz = gl.matrix.nnz()
alg_bytes = nnz * 16 + n * F * 8
T = A.TauMode

def run(name, iters=5):
    res = None
    for it in range(iters):
        if it == 1:
            ctx.profile(True)
        res = compute_taumode_signals(Xd, gl.matrix, T.Fixed(0.5), ctx=ctx)
        ctx.synchronize()
    prof = ctx.profile_all()
    ctx.profile(False)
    ms = prof.get("signal_sums", (0, 1))
    t = ms[0] / ms[1]
    print("  %-44s sums %.3f ms  %.2f TB/s algorithmic (%.3f of 8 TB/s)" % (name, t, alg_bytes / t / 1e9, alg_bytes / t / 1e9 / 8.0), flush=True)
    return res.cpu().numpy()

orders = {"the join's own": "keep", "by cluster (i mod 256)": np.argsort(np.arange(n) % 256, kind="stable").astype(np.uint32)}
for oname, order in orders.items():
    for ch in (8, 16, 32):
        got = []
        for half in (0, 1):
            ctx.set_option("signal_chunk_rows", ch)
            ctx.set_option("signal_half_slabs", half)
            if isinstance(order, str):
                if ch != 16 or half:
                    pass
                # the stream is rebuilt when an order is set; the join's order is re-attached by rebuilding the graph
                gl = A.build_laplacian_matrix(Xd, gp, ctx=ctx)
            else:
                gl.matrix.set_order(order)
            got.append(run("%s chunk=%d half=%d" % (oname, ch, half)))
        assert np.array_equal(got[0], got[1]), "half slabs changed the result"
print("half slabs: bit-identical lambdas")
