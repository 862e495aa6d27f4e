"""Where the masked f16 join's time goes at n x 128: the product kernel, the diagnostic instantiation with no bit that changes
anything (bit 30), without the flush's atomics and stores (16384), with the listing cut short behind its prefix sums (0x8000000), behind the sign-bit
word (0x4000000), behind the pretest (0x2000000), without the listing at all (2).
usage: join_phases.py [n]"""
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
A.build_laplacian_matrix(Xd, gp, ctx=ctx); ctx.synchronize()
for dbg in (0, 1 << 30, 16384, 0x8000000, 0x4000000, 0x2000000, 2, 0):
    ctx.set_option("debug", dbg)
    ts = []
    for it in range(3):
        ctx.profile(True)
        try:
            A.build_laplacian_matrix(Xd, gp, ctx=ctx)
        except Exception as e:  # (a build whose lists are empty may fail later: the join's time is recorded anyway)
            pass
        ctx.synchronize()
        prof = ctx.profile_all(); ctx.profile(False)
        ts.append(prof["knn_mfma_join_f16"][0])
    print("debug=%d: join_f16 %s ms, blocks %d listed %d" % (dbg, ["%.3f" % t for t in ts], ctx.stat("knn_mfma_join_f16_blocks"),
                                                            ctx.stat("knn_mfma_join_f16_blocks_listed")), flush=True)
ctx.set_option("debug", 0)
