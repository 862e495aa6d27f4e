"""The dense f16 join (every bound off) at 1 M x 128: the product kernel, the diagnostic instantiation, without listing (2),
without the coarse test (128), the cycle stamps (8).  ARROWSPACE_ALLOW_ABLATION=1 for the ablation bits."""
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
for opt, val in (("join_prefilter", 0), ("join_cone_tiles", 0), ("join_row_order", 0), ("join_theta_band", 0)):
    ctx.set_option(opt, val)
A.build_laplacian_matrix(Xd, gp, ctx=ctx); ctx.synchronize()
for dbg in (0, 1 << 30, 2, 128, 8, 0):
    ctx.set_option("debug", dbg)
    ts = []
    for it in range(2):
        ctx.profile(True)
        A.build_laplacian_matrix(Xd, gp, ctx=ctx)
        ctx.synchronize()
        prof = ctx.profile_all(); ctx.profile(False)
        ts.append(prof["knn_mfma_join_f16"][0])
    blocks = ctx.stat("knn_mfma_join_f16_blocks")
    print("debug=%d: join_f16 %s ms, blocks %d listed %d -> %.0f TFLOP/s" % (dbg, ["%.2f" % t for t in ts], blocks,
          ctx.stat("knn_mfma_join_f16_blocks_listed"), 2.0 * 32 * 32 * 128 * blocks / (min(ts) * 1e-3) / 1e12), flush=True)
ctx.set_option("debug", 0)
