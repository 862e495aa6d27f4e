"""Wall time of the masked f16 pass under the kernel's ablation bits (outputs undefined): usage ablate_join.py [n]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
Xd, cent = synth.mixture_device(n, 128, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
ctx.set_option("join_grouped_lists", 0)
A.build_laplacian_matrix(Xd, gp, ctx=ctx); ctx.synchronize()
for name, bits in (("full", 0), ("no appends in the flush (16384)", 16384), ("no listing (2)", 2), ("no test, no listing (128)", 128), ("register walk (bit 23)", 0x800000)):
    ctx.set_option("debug", bits)
    for it in range(3):
        if it == 1: ctx.profile(True)
        try:
            A.build_laplacian_matrix(Xd, gp, ctx=ctx)
        except Exception as e:
            pass
        ctx.synchronize()
    prof = ctx.profile_all(); ctx.profile(False)
    ms, cnt = prof.get("knn_mfma_join_f16", (0, 1))
    print("%-40s join_f16 %.3f ms" % (name, ms / max(cnt, 1)), flush=True)
ctx.set_option("debug", 0)
